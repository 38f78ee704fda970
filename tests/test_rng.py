"""glibc rand() model (oracle/glibc_rand.h and csrc GlibcRng) against libc itself."""
import ctypes
import subprocess
import sys

import numpy as np


def _libc_hem_rand(n):
    """n hem::rand() values from the real libc rand() in a FRESH process (state is process-global)."""
    code = ("import ctypes,sys\nlibc=ctypes.CDLL('libc.so.6')\nout=[]\n"
            f"for _ in range({n}):\n r=0\n for i in range(8):\n  r|=(libc.rand()%16)<<(4*i)\n out.append(r)\nprint(' '.join(map(str,out)))")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, check=True)
    return np.array([int(x) for x in r.stdout.split()], dtype=np.uint64).astype(np.uint32)


def test_glibc_model_matches_libc(oracle):
    want = _libc_hem_rand(5000)
    got = oracle.rand_stream(5000)
    assert np.array_equal(got, want)
    # skipping k draws = the tail of the stream
    assert np.array_equal(oracle.rand_stream(100, skip=1234), want[1234:1334])


def test_srand_seed(oracle):
    libc = ctypes.CDLL("libc.so.6")
    libc.srand(12345)
    want = [libc.rand() for _ in range(16)]
    got = oracle.rand_stream(2, seed=12345)
    # first hem::rand packs the low nibbles of the first eight rand() values
    r = 0
    for i in range(8):
        r |= (want[i] % 16) << (4 * i)
    assert int(got[0]) == r


def test_parent_probability_semantics(oracle):
    # rand01 = float(r)/4294967296f ; flag = rand01 < float(1/rho)
    r = oracle.rand_stream(20000)
    flags = oracle.parent_flags(20000, 3.0)
    want = (r.astype(np.float32) / np.float32(4294967296.0)) < (np.float32(1.0) / np.float32(3.0))
    assert np.array_equal(flags.astype(bool), want)
    assert abs(flags.mean() - 1 / 3) < 0.02
