#!/bin/bash
# Same-box A/B of two builds of the library: alternates `python bench.py ARGS` with OFASR_LIB_PATH=<base .so> and the
# in-tree build.  usage: bash tools/ab_lib.sh ab/libofasr_base.so [rounds] [bench.py args]
BASE=$1; ROUNDS=${2:-3}; shift; shift || true
ARGS=${*:---no-cpu-baseline --no-fp32 --no-roofline}
R=${GRAFT_REPO_ROOT:-$(pwd)}
one() { python3 $R/bench.py $ARGS 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('%-5s %8.1f %s  %.3f ms' % (sys.argv[1], d['value'], d['unit'], d['ms_per_step']))" $1; }
for i in $(seq $ROUNDS); do
    OFASR_LIB_PATH=$R/$BASE one base
    one new
done
