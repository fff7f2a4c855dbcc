#!/bin/bash
# Runs one gpurun call in the background of the build container; log in /tmp/gpurun_<tag>.log
# usage: tools/gpu_bg.sh <tag> <timeout_s> '<command>'
tag=$1; to=$2; shift 2
( /usr/local/graft/bin/gpurun --timeout "$to" -- "$@" > /tmp/gpurun_$tag.log 2>&1; echo "__DONE__ rc=$?" >> /tmp/gpurun_$tag.log ) &
