#!/bin/bash
# conv K order A/B: channel-chunk-major (library) vs tap-major (variant): lab table, bench, conv tests, HBM fetch of the step
mkdir -p gpurun_out/r03p
O=$PWD/gpurun_out/r03p
R=$PWD
timeout 600 python -m pytest tests -m gpu -x -q -k "conv or vae or golden or unet_forward" > $O/pytest_conv.log 2>&1; tail -2 $O/pytest_conv.log
build/lab_gemm 20 > $O/lab_new.log 2>&1
LD_LIBRARY_PATH=$R/build/variants/tapmajor build/lab_gemm 20 > $O/lab_tapmajor.log 2>&1
python - <<'PY'
import re
def load(p):
    d={}
    for l in open(p):
        m=re.match(r"(.{28})\s+(\d+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)",l)
        if m: d[m.group(1).strip()]=(int(m.group(2)),float(m.group(3)),float(m.group(5)))
    return d
a=load("gpurun_out/r03p/lab_new.log"); b=load("gpurun_out/r03p/lab_tapmajor.log")
print(f"{'shape':30s} calls  chunk-major us  tap-major us")
for k in a:
    if k in b and ("conv" in k): print(f"{k:30s} {a[k][0]:4d} {a[k][1]:10.1f} {b[k][1]:10.1f}")
print("TOTAL ms/step", sum(v[2] for v in a.values()), sum(v[2] for v in b.values()))
PY
python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-train > $O/bench_new.json 2> $O/bench_new.err; cut -c1-220 $O/bench_new.json
SEER_HIP_LIB=$R/build/variants/libseer_tapmajor.so python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-train > $O/bench_tapmajor.json 2> $O/bench_tapmajor.err; cut -c1-220 $O/bench_tapmajor.json
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_new -- python3 $R/scripts/pmc_step.py > $O/pmc_new.log 2>&1
cd $R
python - <<'PY'
import csv,glob,collections
for tag in ("new",):
    f=glob.glob(f"gpurun_out/r03p/pmc_{tag}/**/*counter_collection.csv",recursive=True)
    if not f: print("no pmc csv"); continue
    tot=collections.defaultdict(lambda:[0,0.0])
    for r in csv.DictReader(open(f[0])):
        if r["Counter_Name"]!="FETCH_SIZE": continue
        n=r["Kernel_Name"]; k="seer_gemm_kernel" if "seer_gemm_kernel" in n else None
        if not k: continue
        conv = ", true," in n.split("seer_gemm_kernel")[1][:40] or "true" in n.split("<")[1].split(",")[2]
        key=k+("<conv>" if conv else "<plain>")
        tot[key][0]+=1; tot[key][1]+=float(r["Counter_Value"])
    for k,(c,v) in tot.items(): print(tag,k,c,"launches", f"{2*v*1024/c/1e6:.1f} MB fetched per launch")
PY
rm -rf $O/pmc_new
