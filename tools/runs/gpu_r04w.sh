#!/bin/bash
# round 4, run w: run p's records (PMC passes, smoke, whole GPU suite, bench line, kernel stats) on the tree that splits 128-tile launches in two
exec bash "$(dirname "$0")/gpu_r04p.sh"
