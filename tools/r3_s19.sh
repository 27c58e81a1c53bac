#!/bin/bash
echo "== full"; timeout 500 python tools/rccl_capi_world1.py 2>&1 | grep "^{\|Error\|error\|VIOLATION" | tail -2 | cut -c1-1200
