# Round-6 records on one box (run via gpurun from the repo root).  Stages by first argument list, default all:
#   tests    the GPU suite's log
#   bench    the ONE default line (C2 headline + configs: C3, C4, C5's share) and the eight-rank launch folded onto this GPU
#   pmc      every BASELINE config's fill under rocprofv3: kernel trace + stats, WRITE_SIZE / FETCH_SIZE in their own passes
#            (C2: every counter group of profiles/collect.sh)
#   e2e      the host-level calls' kernels (durations + SQ_INSTS_VALU) -> e2e_roofline.json
#     bash profiles/scripts/r06_records.sh <tag> [tests bench pmc e2e]
TAG=${1:-r06}; shift
STAGES=${*:-tests bench pmc e2e}
mkdir -p gpurun_out/$TAG
R=${GRAFT_REPO_ROOT:-$PWD}
for s in $STAGES; do
case $s in
tests)
  python -m pytest tests -m gpu -q > gpurun_out/$TAG/${TAG}_gputests.log 2>&1; echo "pytest rc $?" >> gpurun_out/$TAG/${TAG}_gputests.log
  grep -v amdgpu.ids gpurun_out/$TAG/${TAG}_gputests.log | tail -4 ;;
bench)
  T0=$SECONDS; python bench.py > gpurun_out/$TAG/${TAG}_bench_default.json 2> gpurun_out/$TAG/bench_default.err
  echo "bench default wall $((SECONDS - T0)) s"; grep -v amdgpu.ids gpurun_out/$TAG/bench_default.err | tail -3
  python - $TAG <<'PY'
import json, sys
tag = sys.argv[1]
r = json.load(open(f"gpurun_out/{tag}/{tag}_bench_default.json"))
def show(n, r):
    if "error" in r:
        print(n, "FAILED", r["error"]); return
    e, rf = r.get("e2e", {}), r["roofline"]
    print(n, "value %.1f frac %.3f unplaced %.3f kernel_ms %.4f" % (r["value"], rf["frac"], rf.get("frac_unplaced") or 0, rf["kernel_ms"]),
          "e2e ms %.3f first %.1f cold %.3f" % (e.get("ms", 0), e.get("first_call_ms", 0), e.get("cold_ms", 0)),
          "4hits", e.get("up_to_4_hits", {}).get("ms"), "stream", (e.get("stream") or {}).get("value"), "swred", r.get("sw_reduce", {}).get("frac"),
          "walk s", (r.get("arena_placement_search") or r.get("config", {}).get("arena_placement_search") or {}).get("seconds"), "spent", r.get("seconds_spent"))
show("C2", r)
for n, c in r.get("configs", {}).items():
    show(n, c)
c = r.get("cpu_baseline")
if c:
    print("  cpu", c["value"], c["all_cores"]["value"], c["all_cores"]["threads"], c["all_cores"]["scaling_vs_one_thread"], "e2e", c["e2e"]["value"], c["e2e"]["all_cores"]["value"])
PY
  python bench.py --gpus 8 --pairs 8000 --steps 3 --warmup 1 > gpurun_out/$TAG/${TAG}_bench_n8_folded.json 2> gpurun_out/$TAG/bench_n8.err
  head -c 300 gpurun_out/$TAG/${TAG}_bench_n8_folded.json; echo ;;
pmc)
  cd $R
  bash profiles/collect.sh ${TAG}_C2 > gpurun_out/$TAG/collect_C2.log 2>&1
  for w in C3 C4 C5; do PMC_SETS=traffic bash profiles/collect.sh ${TAG}_$w --workload $w > gpurun_out/$TAG/collect_$w.log 2>&1; done
  for w in C2 C3 C4 C5; do
    cp gpurun_out/prof_${TAG}_$w/summary.json gpurun_out/$TAG/${TAG}_$w.json; cp gpurun_out/prof_${TAG}_$w/kernel_stats.csv gpurun_out/$TAG/${TAG}_${w}_kernel_stats.csv
    grep -h '"metric"' gpurun_out/prof_${TAG}_$w/stats.log | sed 's/^[^{]*//' > gpurun_out/$TAG/${TAG}_${w}_bench_line_under_rocprof.json
    python - gpurun_out/$TAG/${TAG}_$w.json $w <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
for k, v in d.get("pmc_per_launch", {}).items():
    if k.startswith("fill_stream"):
        print(sys.argv[2], k[:48], "write %.4f GB fetch %.1f MB" % (v.get("write_bytes", 0) / 1e9, v.get("fetch_bytes_corrected", 0) / 1e6),
              "trace us", {kk: round(vv["mean"], 1) for kk, vv in d.get("kernel_trace_us", {}).items() if kk.startswith("fill_stream")})
PY
  done ;;
e2e)
  cd $R
  bash profiles/scripts/e2e_roofline.sh $TAG > gpurun_out/$TAG/e2e_roofline.log 2>&1
  tail -3 gpurun_out/$TAG/e2e_roofline.log ;;
esac
done
