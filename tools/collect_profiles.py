"""Copies the summaries of gpurun_out/final_r2 (tools/refresh_profiles.sh) into profiles/ under round-2 names."""
import glob, os, shutil
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(R, "gpurun_out", "final_r2"); dst = os.path.join(R, "profiles")
pairs = [("bench.json", "r2_bench.json"), ("r2_pmc.json", "r2_pmc.json")]
for tag in ("cornell", "bunny_cornell"):
    pairs += [(f"bench_under_rocprof_{tag}.json", f"r2_{tag}_bench_under_rocprof.json"),
              (f"bench_under_rocprof_inflight1_{tag}.json", f"r2_{tag}_bench_under_rocprof_inflight1.json")]
    for d, name in ((f"stats_{tag}", f"r2_{tag}_kernel_stats.csv"), (f"stats1_{tag}", f"r2_{tag}_kernel_stats_inflight1.csv")):
        f = glob.glob(os.path.join(src, d, "**", "*kernel_stats.csv"), recursive=True)
        if f:
            shutil.copy(f[0], os.path.join(dst, name)); print("copied", name)
for a, b in pairs:
    if os.path.exists(os.path.join(src, a)):
        shutil.copy(os.path.join(src, a), os.path.join(dst, b)); print("copied", b)
