"""Copies the summaries of gpurun_out/final_r3 (tools/refresh_profiles.sh) into profiles/ under round-3 names."""
import glob, os, shutil
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(R, "gpurun_out", "final_r3"); dst = os.path.join(R, "profiles")
pairs = [("bench.json", "r3_bench.json"), ("bench_driver_style.json", "r3_bench_driver_style.json"), ("r3_pmc.json", "r3_pmc.json"), ("issue_cost.txt", "r3_issue_cost.txt")]
for tag in ("cornell", "bunny_cornell"):
    pairs += [(f"bench_under_rocprof_{tag}.json", f"r3_{tag}_bench_under_rocprof.json")]
    stats = [(f"stats_{tag}", f"r3_{tag}_kernel_stats.csv")]
    for p in (0, 8):
        pairs += [(f"bench_under_rocprof_inflight1_{tag}_p{p}.json", f"r3_{tag}_bench_under_rocprof_inflight1_p{p}.json")]
        stats += [(f"stats1_{tag}_p{p}", f"r3_{tag}_kernel_stats_inflight1_p{p}.csv")]
    for d, name in stats:
        f = glob.glob(os.path.join(src, d, "**", "*kernel_stats.csv"), recursive=True)
        if f:
            shutil.copy(f[0], os.path.join(dst, name)); print("copied", name)
for a, b in pairs:
    if os.path.exists(os.path.join(src, a)):
        shutil.copy(os.path.join(src, a), os.path.join(dst, b)); print("copied", b)
