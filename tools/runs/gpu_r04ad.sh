#!/bin/bash
# round 4, run ad: run p's records (PMC passes, smoke, whole GPU suite, bench line, kernel stats) on the final tree (adds the GEGLU-epilogue instantiation)
exec bash "$(dirname "$0")/gpu_r04p.sh"
