#!/bin/bash
# usage: tools_sweep.sh "<bench args>" word1 word2 ...   prints value/frac per variant word
ARGS=$1; shift
for W in "$@"; do
  R=$(timeout -k 10 200 python bench.py $ARGS --no-cpu-baseline --variant $W 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.1f %.3f' % (d['value'], d['roofline']['frac']))")
  echo "variant $W leave=$(( ((W>>8)&255)-1 )) heavy=$(( ((W>>16)&255)-1 )) leafBias=$(( (W>>24)&255 )) -> $R"
done
