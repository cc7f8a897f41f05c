"""Times the eight reference-named functions one by one (bench.dropin_sequence) at the reference driver's shape and BASELINE config 2's,
copying and resident mode; with `--old DIR` also a module of an earlier revision (DIR/old_gccNMFFunctions.py) on the same box.
    python scripts/dropin_times.py [--old lab_old] > gpurun_out/dropin_times.json"""
import json
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import bench                                                     # noqa: E402

x = bench.dev1_mixture()
out = {}
for name, hop, K in (('driver_shape_hop128_K128', 128, 128), ('config2_hop256_K1024', 256, 1024)):
    out[name] = {'copying': bench.dropin_sequence(x, 16000, hop, K), 'resident': bench.dropin_sequence(x, 16000, hop, K, resident=True)}
if '--old' in sys.argv:
    sys.path.insert(0, os.path.join(REPO, sys.argv[sys.argv.index('--old') + 1]))
    import old_gccNMFFunctions as G0
    for name, hop, K in (('driver_shape_hop128_K128', 128, 128), ('config2_hop256_K1024', 256, 1024)):
        out[name]['previous_revision'] = bench.dropin_sequence(x, 16000, hop, K, G=G0)
print(json.dumps(out, indent=1))
