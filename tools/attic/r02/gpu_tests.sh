#!/bin/bash
# the GPU suite (optionally a subset: args are passed to pytest)
export TMPDIR=/tmp
if [ $# -gt 0 ]; then timeout 2400 python -m pytest "$@" -x -q -m gpu 2>&1 | tail -25
else timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -25; fi
