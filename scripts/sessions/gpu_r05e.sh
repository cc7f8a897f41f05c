#!/bin/bash
# Why is K3 launched back to back slower in the new kernel when the launch has more than one round of workgroups?  Timelines, both libraries.
TAG=${1:-r05e}
OUT=gpurun_out/$TAG; export OUT
mkdir -p $OUT
export TMPDIR=/tmp
OLD=$PWD/gcc_nmf_amd/libgccnmf_hip_r04.so
for f in 51 64; do
  for rep in 0 3; do
    timeout 300 python scripts/ktrace.py --files $f --stage 3 --repeat $rep > $OUT/ktrace_new_f${f}_r${rep}.txt 2>&1; echo "new f$f r$rep exit $?"; head -4 $OUT/ktrace_new_f${f}_r${rep}.txt
    GCCNMF_HIP_LIB=$OLD timeout 300 python scripts/ktrace.py --files $f --stage 3 --repeat $rep > $OUT/ktrace_r04_f${f}_r${rep}.txt 2>&1; echo "r04 f$f r$rep exit $?"; head -4 $OUT/ktrace_r04_f${f}_r${rep}.txt
  done
done
python - <<'PY'
# event timing: stage 3 alone vs back to back, both libraries, in separate processes
import subprocess, os, sys
code = r'''
import sys, os, torch, numpy as np
sys.path.insert(0, os.getcwd())
from gcc_nmf_amd import _hip
from gcc_nmf_amd.engine import Geometry, _ptr, _stream
lib = _hip.lib()
for B in (32, 51, 64):
    F, T, K = 513, 622, 1024
    g = Geometry(F, T, K); N = g.N
    gen = torch.Generator(device='cuda').manual_seed(0)
    V = torch.zeros((B, g.Fp, g.Np), device='cuda'); W = torch.zeros((B, g.Fp, g.Kp), device='cuda'); H = torch.zeros((B, g.Kp, g.Np), device='cuda')
    V[:, :F, :N] = torch.rand((B, F, N), device='cuda', generator=gen) + 0.01
    W[:, :F, :K] = torch.rand((B, F, K), device='cuda', generator=gen) + 0.01
    H[:, :K, :N] = torch.rand((B, K, N), device='cuda', generator=gen) + 0.01
    ws = torch.zeros(lib.gccnmf_klnmf_workspace_floats(F, N, K, B), device='cuda')
    def stage(s): lib.gccnmf_klnmf_stage(_ptr(V), _ptr(W), _ptr(H), _ptr(ws), F, N, K, B, 0.0, 1e-16, 0, s, _stream())
    stage(0)
    for s in range(1, 6): stage(s)
    def timed(seq, reps=10):
        ts = []
        for _ in range(reps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize(); e0.record()
            for s in seq: stage(s)
            e1.record(); e1.synchronize(); ts.append(e0.elapsed_time(e1))
        return float(np.median(ts))
    one = timed([3]); ten = timed([3] * 10) / 10; it = timed([1, 2, 3, 4, 5] * 4) / 4; k1k3 = timed([1, 3] * 5) / 10; k2k3 = (timed([2, 3] * 5) / 5)
    print('files %d: K3 alone %.4f  K3 x10 back to back %.4f each  K1,K3 alternating %.4f each  K2+K3 pair %.4f  iteration %.4f' % (B, one, ten, k1k3, k2k3, it), flush=True)
'''
for name, env in (('r04', {'GCCNMF_HIP_LIB': os.environ.get('PWD', '.') + '/gcc_nmf_amd/libgccnmf_hip_r04.so'}), ('new', {})):
    e = dict(os.environ); e.update(env)
    r = subprocess.run([sys.executable, '-c', code], env=e, capture_output=True, text=True, timeout=600)
    print('==', name); print(r.stdout); print(r.stderr[-500:] if r.returncode else '')
PY
