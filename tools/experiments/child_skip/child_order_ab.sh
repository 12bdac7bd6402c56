#!/bin/bash
# A/B of the children-level tile orders on the bench frame: per-kernel average launch times from bench.py --detail
mkdir -p gpurun_out/r3
run() { # name, env...
  name=$1; shift
  env "$@" python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --serving-frames 0 --detail gpurun_out/r3/ab_$name.detail.json > gpurun_out/r3/ab_$name.json 2> gpurun_out/r3/ab_$name.err
}
run canon_noskip PCGC_CHILD_SORT_MIN=-1 PCGC_CHILD_NOSKIP=1
run canon PCGC_CHILD_SORT_MIN=-1
run sorted PCGC_CHILD_SORT_CHUNK=0
run chunk16k PCGC_CHILD_SORT_CHUNK=16384
run chunk4k PCGC_CHILD_SORT_CHUNK=4096
run chunk1k PCGC_CHILD_SORT_CHUNK=1024
python3 - <<'PY'
import json
names=['canon_noskip','canon','sorted','chunk16k','chunk4k','chunk1k']
rows={}
for n in names:
    try:
        d=json.load(open(f'gpurun_out/r3/ab_{n}.json')); det=json.load(open(f'gpurun_out/r3/ab_{n}.detail.json'))
    except Exception as e:
        print(n, 'failed', e); continue
    print(f"{n:14s} value {d['value']:.1f} ms {d['ms_per_step']:.3f} enc {d['config']['enc_ms']:.2f} dec {d['config']['dec_ms']:.2f}")
    for r in det:
        if 'child' in r['kernel']:
            rows.setdefault((r['kernel'].split(' (')[0], r['n_out']), {})[n]=r['avg_us']
for k,v in sorted(rows.items(), key=lambda kv: -kv[0][1]):
    print(f"{k[0]:26s} {k[1]:8d} " + ' '.join(f"{v.get(n,0):7.1f}" for n in names))
PY
