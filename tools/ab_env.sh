R=${GRAFT_REPO_ROOT:-$(pwd)}
one() { python3 $R/bench.py --no-cpu-baseline --no-fp32 --no-roofline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('%-28s %8.1f  %.3f ms' % (sys.argv[1], d['value'], d['ms_per_step']))" "$1"; }
for i in 1 2; do
  one default
  OFASR_MBCONV_BN_BWD_STAT=1 one BN_BWD_STAT=1
  OFASR_MBCONV_WG1_BX=1 one WG1_BX=1
  OFASR_BN_BWD_ONEPASS=1 one BN_BWD_ONEPASS=1
done
