# Round-5 records on one box (run via gpurun from the repo root): the GPU suite's log, the four bench lines, the e2e rooflines.
TAG=${1:-r05}
mkdir -p gpurun_out/$TAG
python -m pytest tests -m gpu -q > gpurun_out/$TAG/${TAG}_gputests.log 2>&1; echo "pytest rc $?" >> gpurun_out/$TAG/${TAG}_gputests.log
grep -v amdgpu.ids gpurun_out/$TAG/${TAG}_gputests.log | tail -3
python bench.py > gpurun_out/$TAG/${TAG}_bench_default.json 2> gpurun_out/$TAG/bench_default.err
python bench.py --workload C5 --steps 20 > gpurun_out/$TAG/${TAG}_bench_C5.json 2> gpurun_out/$TAG/bench_C5.err
python bench.py --workload C3 --steps 50 > gpurun_out/$TAG/${TAG}_bench_C3.json 2> gpurun_out/$TAG/bench_C3.err
python bench.py --workload C4 --steps 100 > gpurun_out/$TAG/${TAG}_bench_C4.json 2> gpurun_out/$TAG/bench_C4.err
python - $TAG <<'PY'
import json, sys
tag = sys.argv[1]
for n in ("default", "C5", "C3", "C4"):
    try:
        r = json.load(open(f"gpurun_out/{tag}/{tag}_bench_{n}.json"))
        e = r.get("e2e", {})
        print(n, "value %.1f frac %.3f unplaced %.3f" % (r["value"], r["roofline"]["frac"], r["roofline"].get("frac_unplaced") or 0),
              "e2e ms %.3f first %.1f cold %.3f" % (e.get("ms", 0), e.get("first_call_ms", 0), e.get("cold_ms", 0)),
              "4hits", e.get("up_to_4_hits", {}).get("ms"), "swred", r.get("sw_reduce", {}).get("frac"))
        if "cpu_baseline" in r:
            c = r["cpu_baseline"]
            print("  cpu", c["value"], c["all_cores"]["value"], c["all_cores"]["threads"], c["all_cores"]["scaling_vs_one_thread"], "e2e", c["e2e"]["value"], c["e2e"]["all_cores"]["value"])
    except Exception as ex:
        print(n, "FAILED", ex)
PY
bash profiles/scripts/e2e_roofline.sh $TAG > gpurun_out/$TAG/e2e_roofline.log 2>&1
tail -3 gpurun_out/$TAG/e2e_roofline.log
# the headline kernel under rocprofv3 (kernel trace + the PMC passes, each its own run) and the eight-rank launch folded onto this box's GPU
cd $GRAFT_REPO_ROOT
bash profiles/collect.sh ${TAG}_C2 > gpurun_out/$TAG/collect_C2.log 2>&1
cp gpurun_out/prof_${TAG}_C2/summary.json gpurun_out/$TAG/${TAG}_C2.json; cp gpurun_out/prof_${TAG}_C2/kernel_stats.csv gpurun_out/$TAG/${TAG}_C2_kernel_stats.csv
head -c 600 gpurun_out/$TAG/${TAG}_C2.json; echo
python bench.py --gpus 8 --pairs 8000 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/$TAG/${TAG}_bench_n8_folded.json 2> gpurun_out/$TAG/bench_n8.err
head -c 400 gpurun_out/$TAG/${TAG}_bench_n8_folded.json; echo
