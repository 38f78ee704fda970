"""Where does a k_select wave spend its clock?  Needs a variant library built with -DGSR_SELECT_PROFILE
(scripts/build_variant_flags.sh selprof -DGSR_SELECT_PROFILE; GSR_HIP_LIB=variants/selprof.so): s_memtime deltas per phase, summed over
all parents of ONE level-1 launch.  usage: GSR_HIP_LIB=$PWD/variants/selprof.so python scripts/select_profile.py [n] [shape ...]"""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gaussiansplattingregistration_amd import hem, synth, _lib

n = int(sys.argv[1]) if len(sys.argv) > 1 else 5_000_000
shapes = sys.argv[2:] or ["iso", "aniso"]
L = _lib.load()
prof = C.CDLL(os.environ["GSR_HIP_LIB"]).gsr_debug_select_profile
prof.argtypes = [C.POINTER(C.c_ulonglong), C.c_int32]
for shape in shapes:
    c = synth.make_cloud_torch(n, seed=100, shape=shape)
    m = hem.HemMixture()
    for rep in range(2):
        m.set_rng("glibc", 1, 0)
        m.set_level0(c["xyz"], c["color"], c["opacity"], c["cov6"], c["sh"])
        prof(None, 1)
        m.run_level()
        st = m.stats()
        out = (C.c_ulonglong * 16)()
        prof(out, 0)
    P = out[7]
    tot = out[0]
    print(f"{shape}: parents profiled {P} (of {st['parents']}), k_select {st['ms_k_select']:.3f} ms, candidates/parent {st['candidates'] / st['parents']:.1f}, pairs/parent {st['pairs'] / st['parents']:.2f}")
    print(f"   clock per parent (s_memtime ticks, 100 MHz): total {tot / P:.1f}  rows {out[1] / P:.1f} ({100 * out[1] / tot:.1f} %)  stage 2 (+ full stage-3 batches) {out[2] / P:.1f} ({100 * out[2] / tot:.1f} %)  "
          f"final stage 3 {out[3] / P:.1f} ({100 * out[3] / tot:.1f} %)  stage-1 stream (loads + filter + queue) {out[4] / P:.1f} ({100 * out[4] / tot:.1f} %)  "
          f"the rest (record load, set-up, masks, the final partial stage-2 batch is in stage 2) {(tot - out[1] - out[2] - out[3] - out[4]) / P:.1f} ({100 * (tot - out[1] - out[2] - out[3] - out[4]) / tot:.1f} %)")
    print(f"   wave prologue {out[5] / P:.1f}, wave flush {out[6] / P:.1f} (per parent; not part of the total above)")
    print(f"   per parent: grid rows {out[10] / P:.1f}, row batches {out[11] / P:.2f}, non-empty rows {out[12] / P:.1f}, flat candidates {out[13] / P:.1f}, chunk groups (192 candidates) {out[14] / P:.2f}, "
          f"stage-2 batches {out[8] / P:.2f} with {out[9] / max(1, out[8]):.1f} survivors each ({out[9] / P:.1f} survivors per parent), accepted {out[15] / P:.2f}")
    del c, m
