# N processes in a row on one box: what placement each finds (quality, kernel roofline) and how its walks went
N=${1:-6}
for i in $(seq 1 $N); do
python bench.py --no-cpu-baseline --no-e2e --no-unplaced --no-configs --steps 100 2>/dev/null | python -c "
import sys, json
r = json.loads(sys.stdin.readline())
s = r['config']['arena_placement_search']
q = s['try_quality']; w2 = s['second_walk_from']
print('proc $i: frac %.3f quality %.3f balanced %.3f tries %d depthB %.0f depthA %.0f | walk1 max %.3f | rest: %s' % (r['roofline']['frac'], s['quality'], s.get('balanced_quality', 0), s['tries'], s['depth_gib'], s['depth_a_gib'], max(q[:w2]) if w2 else -1, ' '.join('%.3f' % x for x in q[w2:])))
"
done
