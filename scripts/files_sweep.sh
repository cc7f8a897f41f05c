# K3 launch time and the one-stream iteration against the batch size (bench.py --files N --nmf-groups 1): an XCD runs total_tiles / 8 of the
# 20-tiles-per-file list on its 64 workgroup slots.  TUNE="21=8" (or any --tune pairs, space separated) sweeps another launch form.
ARGS=""; for t in $TUNE; do ARGS="$ARGS --tune $t"; done
for f in ${FILES:-16 25 26 40 48 51 52 64 76 77 80 96 102}; do
python bench.py --gpus 1 --steps 2 --warmup 1 --files $f --skip-extras --nmf-groups 1 $ARGS 2>/dev/null | python -c "
import json,sys; b=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=b['roofline']; k3=r.get('k3_alone', r); print('files %3d: %4d workgroups = %.3f rounds  K3 alone %.4f ms (%.3f)  %s   one-stream iteration %.3f ms (%.3f of peak)  e2e %.0f frames/s' % ($f, $f*20, $f*20/512.0, k3['avg_launch_ms'], k3['frac'], 'chained call %.3f' % r['frac'] if 'k3_alone' in r else 'plain launches', b['nmf_iteration_one_stream']['ms'], b['nmf_iteration_one_stream']['frac_of_peak'], b['value']))"
done
