#!/bin/bash
# per-dispatch timeline of config-2 steps at eps 1e-3 / 1e-4 / 1e-2 (tools/time_mle_eps.py under rocprofv3 --kernel-trace)
OUT=${OUT:-gpurun_out/epst}; mkdir -p $OUT; export TMPDIR=/tmp; PD=$(mktemp -d /tmp/prof_XXXXXX)
(cd /tmp && rocprofv3 --kernel-trace -d $PD -- python3 $OLDPWD/tools/time_mle_eps.py > $PD.log 2>&1)
python3 tools/rocprof_calls.py $PD > $OUT/eps_calls_all.txt 2>&1
grep -v "at::\|rocclr" $OUT/eps_calls_all.txt | cut -c1-150 > $OUT/eps_calls.txt
python3 - $OUT/eps_calls.txt <<'PY'
import sys, re, collections
rows=[l for l in open(sys.argv[1])]
# steps are delimited by the first identify scan of each step: summarise the strict / iterate durations per step
steps=[]; cur=None
for l in rows:
    m=re.match(r"\s*([\d.]+) us\s+([\d.]+) us\s+(\d+)\s+(.*)", l)
    if not m: continue
    t,d,g,name=float(m.group(1)),float(m.group(2)),int(m.group(3)),m.group(4)
    if "identify_scan" in name and (cur is None or "locs_from_fits" in cur["last"]):
        cur={"t0":t,"k":collections.defaultdict(list),"last":""}; steps.append(cur)
    if cur is None: continue
    key="strict" if "mle_strict" in name else ("iterate" if "g8_iterate" in name else ("init" if "g8_init" in name else ("final" if "g8_final" in name else ("scan" if "identify_scan" in name else None))))
    if key: cur["k"][key].append(round(d,1))
    cur["last"]=name; cur["t1"]=t+d
for i in (10, 40, 70, 100, 130):
    if i < len(steps):
        s=steps[i]; print(f"step {i}: {s['t1']-s['t0']:.0f} us", dict(s["k"]))
PY
