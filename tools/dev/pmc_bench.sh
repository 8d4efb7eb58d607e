#!/bin/bash
# HBM traffic of the dominant kernel during the real bench step (separate --pmc passes; see MI355X_MICROARCH.md HBM).
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/pmc
for set in "FETCH_SIZE" "WRITE_SIZE"; do
  rm -rf /tmp/pmcb_$set
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d /tmp/pmcb_$set -o p -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-forward-test --no-extras > /tmp/pmcb_$set.log 2>&1
done
python - <<'PY'
import csv, glob, json, collections
res = {}
for cname in ["FETCH_SIZE", "WRITE_SIZE"]:
    f = glob.glob(f"/tmp/pmcb_{cname}/*counter_collection.csv")
    agg = collections.defaultdict(lambda: [0.0, set()])
    for r in csv.DictReader(open(f[0])):
        k = r["Kernel_Name"]
        key = ("gemm_nt" if "gemm_nt_kernel" in k else "wgrad_x" if "wgrad_x_kernel" in k else
               "gemm_tn" if "gemm_tn_kernel" in k else "attn_fwd" if "attn_fwd_kernel" in k else
               "attn_bwd" if "attn_bwd" in k else None)   # all variants of each
        if key is None or r["Counter_Name"] != cname: continue
        agg[key][0] += float(r["Counter_Value"]); agg[key][1].add(r["Dispatch_Id"])
    res[cname] = {k: (v[0], len(v[1])) for k, v in agg.items()}
out = {}
for k in res["FETCH_SIZE"]:
    fs, n = res["FETCH_SIZE"][k]
    ws, n2 = res["WRITE_SIZE"].get(k, (0.0, n))
    # counters are in KiB; gfx950 FETCH_SIZE under-reports wide coalesced reads by 2x (MI355X_MICROARCH.md, HBM)
    out[k] = dict(launches=n, fetch_KiB_raw_per_launch=fs / n, write_KiB_per_launch=ws / max(n2, 1),
                  hbm_bytes_per_launch=(2.0 * fs / n + ws / max(n2, 1)) * 1024.0)
import hashlib, time
out["gemm_hip_sha256"] = hashlib.sha256(open("simvg_amd/csrc/gemm.hip", "rb").read()).hexdigest()[:16]
out["wgrad_hip_sha256"] = hashlib.sha256(open("simvg_amd/csrc/wgrad.hip", "rb").read()).hexdigest()[:16]
out["measured"] = time.strftime("%Y-%m-%dT%H:%MZ", time.gmtime())
json.dump(out, open("gpurun_out/pmc/hbm_traffic.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
