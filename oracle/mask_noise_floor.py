"""Reference-side noise floor of the coefficient masks: the SAME NumPy/OpenBLAS float32 pipeline (the oracle port; its KL-NMF is the
reference's own expression, gccNMFFunctions.py:75-81) run with 1 and with N BLAS threads -- only the sgemm summation order differs.
Test infrastructure (CPU, ~5 min on 8 cores): python oracle/mask_noise_floor.py -> tests/golden/mask_noise_floor.json"""
import sys, json, os, numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from threadpoolctl import threadpool_limits
from oracle import gccnmf_oracle as O
from gcc_nmf_amd import wavfile
GOLD = os.path.join(REPO, 'tests', 'golden', 'data')
res={}
for name in sorted(os.listdir(GOLD)):
    if not name.endswith('.wav'): continue
    x, sr = wavfile.wavread(os.path.join(GOLD,name))
    x = np.asarray(x, np.float32)
    runs=[]
    for th in (1, 8):
        with threadpool_limits(limits=th):
            runs.append(O.runGCCNMF(x, sr, 1024, 256, 128, 1.0, 3, dictionarySize=1024, numIterations=100, return_intermediates=True))
    a,b=runs
    Wd=np.linalg.norm(a['W']-b['W'])/np.linalg.norm(a['W']); Hd=np.linalg.norm(a['H']-b['H'])/np.linalg.norm(a['H'])
    G=a['G'].astype(np.float64)            # (S,K,T)
    srt=np.sort(G,axis=0); gap=(srt[-1]-srt[-2])/np.abs(srt[-1])
    am_a, am_b = np.argmax(a['G'],axis=0), np.argmax(b['G'],axis=0)
    flips = am_a!=am_b
    rms=float(np.sqrt(np.mean((a['y'].astype(np.float64)-b['y'])**2)))
    res[name]=dict(W_rel=float(Wd),H_rel=float(Hd),flips=int(flips.sum()),coeffs=int(flips.size),largest_flipped_gap=float(gap[flips].max()) if flips.any() else 0.0,
                   idx_equal=a['idx']==b['idx'], waveform_rms=rms)
    print(name, res[name], flush=True)
json.dump(res, open(os.path.join(REPO, 'tests', 'golden', 'mask_noise_floor.json'), 'w'), indent=1, sort_keys=True)
