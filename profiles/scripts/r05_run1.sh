mkdir -p gpurun_out/r05
python -m pytest tests -m gpu -x -q > gpurun_out/r05/gputests1.log 2>&1; echo "pytest rc $?" >> gpurun_out/r05/gputests1.log
tail -5 gpurun_out/r05/gputests1.log
python bench.py > gpurun_out/r05/bench_default1.json 2> gpurun_out/r05/bench_default1.err; tail -c 3000 gpurun_out/r05/bench_default1.json
python bench.py --workload C5 --steps 20 > gpurun_out/r05/bench_C5_1.json 2> gpurun_out/r05/bench_C5_1.err
python bench.py --workload C3 --steps 50 > gpurun_out/r05/bench_C3_1.json 2> gpurun_out/r05/bench_C3_1.err
python bench.py --workload C4 --steps 100 > gpurun_out/r05/bench_C4_1.json 2> gpurun_out/r05/bench_C4_1.err
python - <<'PY'
import json
for n in ("default1","C5_1","C3_1","C4_1"):
    try:
        r=json.load(open(f"gpurun_out/r05/bench_{n}.json"))
        e=r.get("e2e",{})
        print(n, "value %.1f frac %.3f unplaced %s" % (r["value"], r["roofline"]["frac"], r["roofline"].get("frac_unplaced")), "e2e ms %.3f first %.1f cold %.3f" % (e.get("ms",0), e.get("first_call_ms",0), e.get("cold_ms",0)), "4hits", e.get("up_to_4_hits",{}).get("ms"), "kept", r["config"]["arena_placement_search"].get("kept_gib"), "swred", r.get("sw_reduce",{}).get("frac"))
        if "cpu_baseline" in r:
            c=r["cpu_baseline"]; print("  cpu", c["value"], c["all_cores"]["value"], c["all_cores"]["threads"], c["all_cores"]["scaling_vs_one_thread"], "e2e", c["e2e"]["value"], c["e2e"]["all_cores"]["value"])
    except Exception as ex:
        print(n, "FAILED", ex)
PY
bash profiles/scripts/sweep_pmc.sh r05a C3 > gpurun_out/r05/sweep_pmc_C3.log 2>&1
bash profiles/scripts/sweep_pmc.sh r05a C4 > gpurun_out/r05/sweep_pmc_C4.log 2>&1
