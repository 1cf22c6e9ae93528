#!/bin/bash
# round 4, run z: run p's records (PMC passes, smoke, whole GPU suite, bench line, kernel stats) on the final tree (128-tile split + XCD map 4)
exec bash "$(dirname "$0")/gpu_r04p.sh"
