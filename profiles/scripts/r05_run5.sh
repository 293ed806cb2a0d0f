mkdir -p gpurun_out/r05
python -m pytest tests -m gpu -x -q > gpurun_out/r05/gputests5.log 2>&1; echo "pytest rc $?" >> gpurun_out/r05/gputests5.log
tail -4 gpurun_out/r05/gputests5.log
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for h in 1 4; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r05/c4prof_$h -o t -- python $R/seq-align_amd/tools/sw_enum_profile.py C4 $h > $R/gpurun_out/r05/c4prof_$h.log 2>&1
  grep "max_hits" $R/gpurun_out/r05/c4prof_$h.log
  find $R/gpurun_out/r05/c4prof_$h -name "*kernel_stats.csv" -exec head -5 {} \;
done
cd $R
for wl in C3 C4; do
python bench.py --workload $wl --steps 50 --no-cpu-baseline > gpurun_out/r05/bench_${wl}_5.json 2> gpurun_out/r05/bench_${wl}_5.err
done
python - <<'PY'
import json
for n in ("C3_5","C4_5"):
    try:
        r=json.load(open(f"gpurun_out/r05/bench_{n}.json"))
        e=r.get("e2e",{})
        print(n, "value %.1f frac %.3f" % (r["value"], r["roofline"]["frac"]), "e2e ms %.3f first %.1f cold %.3f" % (e.get("ms",0), e.get("first_call_ms",0), e.get("cold_ms",0)), "4hits", e.get("up_to_4_hits",{}).get("ms"), "swred", r.get("sw_reduce",{}).get("frac"), r.get("sw_reduce",{}).get("kernel_ms"))
    except Exception as ex:
        print(n, "FAILED", ex)
PY
