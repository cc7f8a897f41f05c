#!/bin/bash
export TMPDIR=/tmp
for t in "1=0" "1=64"; do
echo "== tune $t"
GCCNMF_TUNE=$t timeout 300 python scripts/direct_bench.py 1024 256 1 2>&1 | grep -E "^direct "
done
