#!/bin/bash
# HBM traffic of the hot kernels during the real bench step (separate --pmc passes, --kernel-trace only; see
# MI355X_MICROARCH.md HBM: FETCH_SIZE x 2 for wide coalesced reads on gfx950, counters in KiB).  Writes
# gpurun_out/pmc/hbm_traffic.json in the schema of profiles/gemm_nt_hbm_traffic.json (copy it there and commit): the dominant
# kernel (gemm_nt) at the top level, every other family under other_kernels_hbm_bytes_per_launch, and the sha256 of the csrc
# file each family was measured on -- bench.py reports a `traffic` figure only while that hash matches the built source.
# Call it with SIMVG_COMMIT=$(git rev-parse --short HEAD) in the environment (the GPU box has no .git): the JSON is stamped with it.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/pmc
for set in "FETCH_SIZE" "WRITE_SIZE"; do
  rm -rf /tmp/pmcb_$set
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d /tmp/pmcb_$set -o p -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-forward-test --no-extras > /tmp/pmcb_$set.log 2>&1
done
python - <<'PY'
import csv, glob, json, collections, hashlib, os, subprocess, time
FAMILIES = [("gemm_nt", "gemm_nt_kernel", "gemm.hip"), ("wgrad_x", "wgrad_x_kernel", "wgrad.hip"), ("wgrad_x", "wgrad_sq_kernel", "wgrad.hip"), ("wgrad_reduce", "wgrad_slab_reduce", "wgrad.hip"), ("gemm_tn", "gemm_tn_kernel", "gemm.hip"),
            ("attn_fwd", "attn_fwd", "attention.hip"), ("attn_bwd", "attn_bwd", "attention_bwd1.hip"),
            ("ln_fwd", "ln_fwd", "layernorm.hip"), ("ln_bwd", "ln_bwd", "layernorm.hip"), ("ln_bwd", "ln_param_reduce", "layernorm.hip"),
            ("adam", "adam_kernel", "optim.hip"), ("adam", "sumsq_kernel", "optim.hip")]
res = {}
for cname in ["FETCH_SIZE", "WRITE_SIZE"]:
    f = glob.glob(f"/tmp/pmcb_{cname}/*counter_collection.csv")
    agg = collections.defaultdict(lambda: [0.0, set()])
    for r in csv.DictReader(open(f[0])):
        k = r["Kernel_Name"]
        key = next((fam for fam, pat, _ in FAMILIES if pat in k), None)
        if key is None or r["Counter_Name"] != cname: continue
        agg[key][0] += float(r["Counter_Value"]); agg[key][1].add(r["Dispatch_Id"])
    res[cname] = {k: (v[0], len(v[1])) for k, v in agg.items()}
fam = {}
for k in res["FETCH_SIZE"]:
    fs, n = res["FETCH_SIZE"][k]
    ws, n2 = res["WRITE_SIZE"].get(k, (0.0, n))
    fam[k] = dict(launches=n, hbm_bytes_per_launch=(2.0 * fs / n + ws / max(n2, 1)) * 1024.0,
                  fetch_x2_bytes=2.0 * fs / n * 1024.0, write_bytes=ws / max(n2, 1) * 1024.0)
sha = lambda f: hashlib.sha256(open(os.path.join("simvg_amd/csrc", f), "rb").read()).hexdigest()[:16]
old = {}
try:
    old = json.load(open("profiles/gemm_nt_hbm_traffic.json"))
except Exception:
    pass
g = fam.pop("gemm_nt")
out = dict(kernel="gemm_nt (all launches of the gemm_nt_kernel_* family during 3 bench steps: the set the bench's roofline object times)",
           source="rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- python bench.py --steps 2 --warmup 1 "
                  "--no-cpu-baseline --no-forward-test --no-extras (tools/dev/pmc_bench.sh)",
           measured=time.strftime("%Y-%m-%dT%H:%MZ", time.gmtime()), commit=os.environ.get("SIMVG_COMMIT", "working tree"),
           gemm_hip_sha256=sha("gemm.hip"), wgrad_hip_sha256=sha("wgrad.hip"), attention_hip_sha256=sha("attention.hip"), attention_bwd1_hip_sha256=sha("attention_bwd1.hip"),
           layernorm_hip_sha256=sha("layernorm.hip"), optim_hip_sha256=sha("optim.hip"),
           fetch_KiB_raw_per_launch=g["fetch_x2_bytes"] / 2048.0, write_KiB_per_launch=g["write_bytes"] / 1024.0,
           correction="gfx950 FETCH_SIZE reports 1/2 of the bytes of wide coalesced reads (MI355X_MICROARCH.md, HBM) -> fetch doubled; "
                      "counters are L2 misses to the fabric (Infinity-Cache hits included): an upper bound of the HBM bytes",
           hbm_bytes_per_launch=g["hbm_bytes_per_launch"], launches=g["launches"],
           algorithmic_bytes_per_launch_avg=old.get("algorithmic_bytes_per_launch_avg"),
           history=dict(old.get("history", {}), **({f"{old.get('round', 'round 5')} ({old.get('measured')}, commit {old.get('commit')})": old.get("hbm_bytes_per_launch")} if old.get("hbm_bytes_per_launch") else {})),
           round=os.environ.get("SIMVG_ROUND", "round 6"),
           other_kernels_hbm_bytes_per_launch=fam)
json.dump(out, open("gpurun_out/pmc/hbm_traffic.json", "w"), indent=1)
print(json.dumps({k: (round(v["hbm_bytes_per_launch"] / 1e6, 1), v["launches"]) for k, v in dict(fam, gemm_nt=g).items()}, indent=1))
PY
