#!/bin/bash
# round 4, run ai: run p's records (PMC passes, smoke, whole GPU suite, bench line, kernel stats) on the final tree (LDS-DMA kernel instantiated per element type: same kernels, 1.7-minute build)
exec bash "$(dirname "$0")/gpu_r04p.sh"
