#!/bin/bash
# Round 6, session i: the whole GPU suite with the chained launch as the default (key 21 = 1), the experiment build's chain tests, the driver's bench command
TAG=${1:-r06i}
OUT=gpurun_out/$TAG; export OUT
mkdir -p $OUT
export TMPDIR=/tmp
EXP=$PWD/gcc_nmf_amd/libgccnmf_hip_exp.so
timeout 1700 python -m pytest tests -q -m gpu --tb=short -p no:cacheprovider --timeout 300 > $OUT/pytest_gpu.log 2>&1; echo "pytest exit $?"; tail -8 $OUT/pytest_gpu.log
GCCNMF_HIP_LIB=$EXP timeout 600 python -m pytest tests/test_gpu_kernels.py -q -m gpu --tb=short -p no:cacheprovider --timeout 120 -k "chained" > $OUT/pytest_chain_exp.log 2>&1; echo "chain tests (experiment build) exit $?"; tail -3 $OUT/pytest_chain_exp.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke exit $?"; tail -1 $OUT/smoke.log
timeout 1200 python bench.py --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err; echo "bench exit $?"; tail -3 $OUT/bench.err
python - <<'PY'
import json, os
b = json.loads(open(os.path.join(os.environ['OUT'], 'bench.json')).read().strip().splitlines()[-1])
print('value', b['value'], 'ms/step', b['ms_per_step'], 'frac', b['roofline']['frac'], 'iter', b['nmf_iteration_one_stream'], 'groups', b['config']['nmf_file_groups_per_gpu'])
print('traffic', b['roofline'].get('traffic'), b['roofline'].get('traffic_source'))
for k in ('single_file', 'dropin_performKLNMF', 'reference_driver_shape', 'sec8d_host_to_host'):
    print(k, {kk: vv for kk, vv in b.get(k, {}).items() if not isinstance(vv, str)})
for k in ('k128_batch', 'k_sweep', 'it200', 'shared_dictionary_n1', 'big_matrix_n80000'):
    print(k, json.dumps(b.get(k))[:400])
d = b.get('dropin_sequence', {})
for shape in ('driver_shape_hop128_K128', 'config2_hop256_K1024'):
    for mode in ('copying', 'resident'):
        r = d.get(shape, {}).get(mode)
        if r: print(shape, mode, 'sum %.2f ms whole %.2f' % (r['sum_of_the_eight_ms'], r['whole_sequence_ms']))
print('cpu', b.get('cpu_baseline', {}).get('value'), b.get('gpu_vs_cpu_waveform_rms'), b.get('gpu_vs_cpu_tdoa_equal'), 'config_lines_seconds', b.get('config_lines_seconds'))
PY
