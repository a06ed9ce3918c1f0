"""Copies the summaries of gpurun_out/final_<round> (tools/refresh_profiles.sh) into profiles/ under <round>_ names.   collect_profiles.py [round]"""
import glob, os, shutil, sys
RD = sys.argv[1] if len(sys.argv) > 1 else "r5"
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(R, "gpurun_out", "final_" + RD); dst = os.path.join(R, "profiles")
pairs = [("bench.json", RD + "_bench.json"), ("bench_driver_style.json", RD + "_bench_driver_style.json"), (RD + "_pmc.json", RD + "_pmc.json"), ("issue_cost.txt", RD + "_issue_cost.txt"),
         ("other_configs.txt", RD + "_other_configs.txt"), (RD + "_pmc_wait.json", RD + "_pmc_wait.json"),
         (RD + "_timeline_cornell.txt", RD + "_timeline_cornell.txt"), (RD + "_timeline_bunny_cornell.txt", RD + "_timeline_bunny_cornell.txt"), (os.path.join("..", "pmc_timed_" + RD, RD + "_pmc_timed.json"), RD + "_pmc_timed.json")]
for tag in ("cornell", "bunny_cornell"):
    pairs += [(f"bench_under_rocprof_{tag}.json", f"{RD}_{tag}_bench_under_rocprof.json")]
    stats = [(f"stats_{tag}", f"{RD}_{tag}_kernel_stats.csv")]
    for p in (0, 8):
        pairs += [(f"bench_under_rocprof_inflight1_{tag}_p{p}.json", f"{RD}_{tag}_bench_under_rocprof_inflight1_p{p}.json")]
        stats += [(f"stats1_{tag}_p{p}", f"{RD}_{tag}_kernel_stats_inflight1_p{p}.csv")]
    for d, name in stats:
        f = glob.glob(os.path.join(src, d, "**", "*kernel_stats.csv"), recursive=True)
        if f:
            shutil.copy(f[0], os.path.join(dst, name)); print("copied", name)
for a, b in pairs:
    if os.path.exists(os.path.join(src, a)):
        shutil.copy(os.path.join(src, a), os.path.join(dst, b)); print("copied", b)
