"""Argument fuzz of every C-ABI entry point WITHOUT a GPU: each call must come back with a status (non-zero for invalid arguments),
never crash, never read out of bounds on the host.  Run in-process by tests/test_abi_cpu.py and, against the AddressSanitizer build of
the library, in a subprocess under LD_PRELOAD=libclang_rt.asan (usage: abi_fuzz.py <path to .so>)."""
import ctypes as C
import importlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def fuzz(lib_path=None):
    L = importlib.import_module("gesture-generation-from-trimodal-context_amd._lib")
    if lib_path is not None:
        L.LIB_PATH = lib_path
        L._lib = None
    lib = L.load()
    scratch = (C.c_float * 4096)()                       # valid HOST memory: a launcher that dereferences a data pointer would still not fault,
    sp = C.cast(scratch, C.c_void_p)                     # one that indexes past a table would (ASan)
    checked, bad = 0, []

    def value(argtype, mode):
        if argtype in (L.P,):
            return None if mode == 0 else sp
        if argtype in (L.I32, L.I64, L.U32):
            return (0, 0, -1, 1)[mode]
        if argtype is L.F32:
            return 0.0
        if argtype is L.WP:
            w = L.Window(sp.value if mode else None, 0, 0, 0, 0, 0, 0, 0, 0, 0)
            return C.byref(w)
        if argtype == C.POINTER(L.NtProblem):
            return (L.NtProblem * 2)() if mode else None
        if argtype == C.POINTER(L.TnProblem):
            return (L.TnProblem * 2)() if mode else None
        if argtype == C.POINTER(L.AeStepArgs):
            return C.byref(L.AeStepArgs()) if mode else None
        if argtype == C.POINTER(L.P):
            return (L.P * 8)() if mode else None
        raise TypeError(argtype)

    for name, argtypes in sorted(L.SIGNATURES.items()):
        fn = getattr(lib, name)
        for mode in (0, 1, 2, 3):                        # nulls + zeros | host pointers + zero sizes | negative sizes | size 1 on zeroed tables
            args = [value(a, mode) for a in argtypes]
            rc = fn(*args)
            assert isinstance(rc, int), name
            if mode in (0, 2) and rc == 0:
                bad.append(f"{name}: accepted {'null / zero' if mode == 0 else 'negative-size'} arguments")
            checked += 1
    assert not bad, bad
    # size queries on degenerate shapes
    for q in ("tg_gemm_tn_ws_floats", "tg_gru_cluster_ws_bytes", "tg_gru_cluster_bwd_ws_bytes"):
        getattr(lib, q)(*([1] * len(getattr(lib, q).argtypes)))
    assert lib.tg_gemm_nt_family(None) == -1
    assert lib.tg_gemm_nt_ext_supported(None) == 0
    assert lib.tg_gemm_nt_kernel_plan(None, 1, None, None) == -1 and lib.tg_gemm_nt_kernel_plan(None, 0, None, None) == -1
    assert lib.tg_set_nt_mover_waves(7) != 0 and lib.tg_set_nt_mover_waves(-2) != 0 and lib.tg_set_nt_mover_waves(-1) == 0
    assert lib.tg_gemm_tn_kernel_plan(None, 1) == -1 and lib.tg_gemm_tn_kernel_plan((L.TnProblem * 2)(), 2) == -1
    zeroed = (L.NtProblem * 2)()
    assert lib.tg_gemm_nt_kernel_plan(zeroed, 2, None, None) == -1 and lib.tg_gemm_nt_kernel_plan(zeroed, 99, None, None) == -1
    assert lib.tg_bn_fused_supported(0, 0, 0) == 0 and lib.tg_bn_fused_supported(4096, 16, 2) == 1
    return checked


if __name__ == "__main__":
    n = fuzz(sys.argv[1] if len(sys.argv) > 1 else None)
    print("abi fuzz ok:", n, "calls")
