#!/bin/bash
# Package power / shader clock / temperatures while bench.py runs whole training iterations (config 2), sampled with rocm-smi every ~0.5 s.
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=gpurun_out/r4_power; mkdir -p $O
timeout -k 10 400 python bench.py --steps 500 --warmup 5 --nat-steps 0 --no-cpu-baseline > $O/bench_long.json 2> $O/bench_long.err &
pid=$!
sleep 10
for i in $(seq 1 40); do rocm-smi --showpower --showclocks --showtemp 2>/dev/null | grep -E "Power \(W\)|sclk|Temperature \(Sensor (junction|memory)" | tr -s ' \t' ' ' | tr '\n' ';'; echo; sleep 0.3; done > $O/iter_samples.txt
wait $pid
python - <<'PY'
import re, json, statistics as st
v=[]
for l in open('gpurun_out/r4_power/iter_samples.txt'):
    m=re.search(r'sclk clock level: \S+ \((\d+)Mhz\).*Package Power \(W\): ([\d.]+)', l); t=re.search(r'junction\) \(C\): ([\d.]+).*memory\) \(C\): ([\d.]+)', l)
    if m and t: v.append((int(m.group(1)), float(m.group(2)), float(t.group(1)), float(t.group(2))))
busy=[x for x in v if x[1] > 800]
d=json.load(open('gpurun_out/r4_power/bench_long.json'))
print(f"bench.py --steps 500: {d['ms_per_step']:.2f} ms per iteration, roofline.frac {d['roofline']['frac']:.3f}")
print(f"{len(busy)} samples while iterating: sclk median {st.median(x[0] for x in busy):.0f} MHz (min {min(x[0] for x in busy)}, max {max(x[0] for x in busy)}), "
      f"package power median {st.median(x[1] for x in busy):.0f} W (min {min(x[1] for x in busy):.0f}, max {max(x[1] for x in busy):.0f}), junction up to {max(x[2] for x in busy):.0f} C, HBM up to {max(x[3] for x in busy):.0f} C")
print("samples (MHz, W):", " ".join(f"{x[0]}/{x[1]:.0f}" for x in busy))
PY
