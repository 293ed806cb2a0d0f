mkdir -p gpurun_out/r05
python -m pytest tests -m gpu -x -q > gpurun_out/r05/gputests3.log 2>&1; echo "pytest rc $?" >> gpurun_out/r05/gputests3.log
tail -4 gpurun_out/r05/gputests3.log
for wl in C3 C4; do
python bench.py --workload $wl --steps 50 --no-cpu-baseline > gpurun_out/r05/bench_${wl}_3.json 2> gpurun_out/r05/bench_${wl}_3.err
done
python - <<'PY'
import json
for n in ("C3_3","C4_3"):
    try:
        r=json.load(open(f"gpurun_out/r05/bench_{n}.json"))
        e=r.get("e2e",{})
        print(n, "value %.1f frac %.3f" % (r["value"], r["roofline"]["frac"]), "e2e ms %.3f first %.1f cold %.3f" % (e.get("ms",0), e.get("first_call_ms",0), e.get("cold_ms",0)), "4hits", e.get("up_to_4_hits",{}).get("ms"), "swred", r.get("sw_reduce",{}).get("frac"), r.get("sw_reduce",{}).get("kernel_ms"))
    except Exception as ex:
        print(n, "FAILED", ex)
PY
python seq-align_amd/tools/reduce_bench.py 2>&1 | grep -v amdgpu.ids | tail -12
