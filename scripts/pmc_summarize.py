"""profiles/r01_pmc.json from gpurun_out/pmc/pmc_raw.json (written by scripts/pmc_run.sh).
HBM bytes per launch = FETCH_SIZE + WRITE_SIZE (KiB -> bytes).  Calibration (1 GiB device copy, the largest
'copy' launch): a streaming read shows up as 0.5 x bytes in FETCH_SIZE (MI355X_MICROARCH.md) while the
narrow random reads of these kernels are 64-B requests counted at face value (TCC_MISS * 64 B ~= FETCH_SIZE),
so no doubling is applied to them; WRITE_SIZE is exact."""
import json, os, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
raw = json.load(open(os.path.join(root, "gpurun_out", "pmc", "pmc_raw.json")))
out = {
    "command": "scripts/pmc_run.sh: rocprofv3 --kernel-trace --pmc <set> -- python scripts/pmc_workload.py 16 ; separate "
               "passes for FETCH_SIZE, WRITE_SIZE, TCC_HIT/MISS and two SQ sets; scripts/pmc_summarize.py",
    "workload": "bench C3 map (2.23M neural points), 100k voxel-sorted scan points, Kc=81, k=8, decoder 4x64; 10 launches of each kernel",
    "units": "FETCH_SIZE / WRITE_SIZE in KiB; SQ_WAVE_CYCLES, SQ_WAIT_* in quad-cycles",
    "calibration": {"what": "1 GiB device copy (largest launch of the copy class)",
                    "FETCH_SIZE_KiB": raw["copy"]["FETCH_SIZE"]["max"], "WRITE_SIZE_KiB": raw["copy"]["WRITE_SIZE"]["max"],
                    "TCC_MISS": raw["copy"]["TCC_MISS_sum"]["max"],
                    "finding": "streaming read: FETCH_SIZE = 0.5 x bytes while TCC_MISS*64 B = bytes; WRITE_SIZE exact. "
                               "Narrow random reads: FETCH_SIZE ~= TCC_MISS*64 B, no doubling applied."},
    "kernels": {k: {c: round(v["mean"], 1) for c, v in raw[k].items()} for k in ("knn", "knn_brick", "gn") if k in raw},
}
hbm = lambda k: int((raw[k]["FETCH_SIZE"]["mean"] + raw[k]["WRITE_SIZE"]["mean"]) * 1024)
out["knn_hbm_bytes_per_launch"] = hbm("knn")
out["knn_brick_hbm_bytes_per_launch"] = hbm("knn_brick")
out["gn_hbm_bytes_per_launch"] = hbm("gn")
out["gn_kernel"] = "gn_accumulate_quad_kernel<64,false,true> (four lanes per query, persistent blocks, split-fp16 decoder)"
out["gn_mfma_busy_cycles_per_launch"] = raw["gn"]["SQ_VALU_MFMA_BUSY_CYCLES"]["mean"]
out["knn_algorithmic_bytes_per_launch"] = 118770000
json.dump(out, open(os.path.join(root, "profiles", "r01_pmc.json"), "w"), indent=1)
print({k: v for k, v in out.items() if k.endswith("per_launch")})
