#!/bin/bash
# same-box A/B of library builds: tools/ab_layers.sh <precision> <windows> <label=lib[,ENV=VAL...]> ...
# prints one column of per-layer microseconds per build (tools/layers.py)
prec=$1; n=$2; shift 2
for spec in "$@"; do
  label=${spec%%=*}; rest=${spec#*=}
  lib=${rest%%,*}; envs=""
  if [[ "$rest" == *,* ]]; then envs=$(echo "${rest#*,}" | tr ',' ' '); fi
  env SOFTSPOKEN_LIB=$PWD/softspoken_amd/$lib $envs timeout -k 10 200 python tools/layers.py $prec $n 2>&1 | awk -v L=$label '/ us /{name=$0; sub(/ +n=.*/, "", name); n=split(name, p, ">/"); u=$0; sub(/.*n= *[0-9]+ +/, "", u); sub(/ us.*/, "", u); print L, p[n], u} /audio-s\/s/{print L, "TOTAL", $0}' > /tmp/ab_$label.txt
done
python3 - "$@" <<'PY'
import sys, collections
labels = [s.split('=')[0] for s in sys.argv[1:]]
rows = collections.OrderedDict()
for L in labels:
    seen = collections.Counter()
    for line in open('/tmp/ab_%s.txt' % L):
        p = line.split()
        if p[1] == 'TOTAL':
            rows.setdefault('TOTAL', {})[L] = ' '.join(p[2:8]); continue
        key = p[1]; seen[key] += 1
        if seen[key] > 1: key += '#%d' % seen[key]
        rows.setdefault(key, {})[L] = p[2]
print('%-28s' % 'layer' + ''.join('%12s' % L for L in labels))
for k, v in rows.items():
    if k == 'TOTAL': continue
    print('%-28s' % k[:28] + ''.join('%12s' % v.get(L, '-') for L in labels))
for L in labels: print(L, rows.get('TOTAL', {}).get(L))
PY
