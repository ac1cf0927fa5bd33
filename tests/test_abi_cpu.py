"""CPU-side checks of the drop-in boundary: the C-ABI library loads and exports every declared symbol, the host
mirror has the reference's state_dict key set, and the product package never touches oracle/ or /root/reference."""
import ctypes
import os
import re
import sys

import pytest

import torch

from conftest import ROOT


def _build_if_needed():
    lib = os.path.join(ROOT, "gesture-generation-from-trimodal-context_amd", "libtrimodal_hip.so")
    if not os.path.exists(lib):
        import __graft_entry__ as g
        g.build()
    return lib


def test_library_exports_every_declared_symbol(pkg):
    lib_path = _build_if_needed()
    header = open(os.path.join(ROOT, "include", "trimodal_hip.h")).read()
    declared = set(re.findall(r"\b(tg_[a-z0-9_]+)\s*\(", header))
    declared.discard("tg_window")
    lib = ctypes.CDLL(lib_path)
    missing = [n for n in sorted(declared) if not hasattr(lib, n)]
    assert not missing, missing
    bound = set(pkg._lib.SIGNATURES) | {"tg_version", "tg_last_error", "tg_gemm_tn_ws_floats", "tg_set_math_mode", "tg_get_math_mode", "tg_set_deterministic", "tg_get_deterministic", "tg_set_tn_workgroup_cap", "tg_get_tn_workgroup_cap",
                                         "tg_gru_cluster_supported", "tg_gru_cluster_ws_bytes", "tg_gru_cluster_bwd_supported",
                                         "tg_gru_cluster_bwd_ws_bytes", "tg_gemm_nt_family", "tg_gemm_nt_ext_supported", "tg_gemm_nt_kernel_plan", "tg_gemm_tn_kernel_plan", "tg_set_nt_mover_waves", "tg_bn_fused_supported", "tg_gru_cluster_fused_dropout", "tg_ae_step_ws_bytes", "tg_ae_step_supported",
                                         "tg_wav_front_ws_doubles", "tg_wav_front_fstat_doubles", "tg_wav_front_gate_words", "tg_bn2_supported", "tg_bn2_ws_doubles", "tg_speaker_bwd_max_rows", "tg_wav_conv2_wgrad_ws_floats", "tg_d_preconv_fwd_supported", "tg_d_preconv_ws_bytes",
                                         "tg_gru_vec_supported", "tg_gru_vec_ws_bytes", "tg_gru_vec_ws_header_bytes"}
    assert declared == bound, declared ^ bound
    lib.tg_version.restype = ctypes.c_int
    assert lib.tg_version() == pkg._lib.ABI_VERSION


def test_argument_checks_fail_loudly_without_gpu(pkg):
    """Entry points validate arguments on the host before any launch: callable here, no GPU needed."""
    lib = pkg._lib.load()
    rc = lib.tg_adam_step(None, None, None, None, 0, 0.1, 0.5, 0.999, 1e-8, None, None)
    assert rc != 0 and b"tg_adam_step" in lib.tg_last_error()
    rc = lib.tg_gru_forward(None, 0, None, None, None, None, None, None, 0, 1, 1, 4, None)
    assert rc != 0


def test_state_dict_keys_match_reference(pkg):
    import argparse
    from oracle import ref_model as O
    a = argparse.Namespace(n_pre_poses=4, n_poses=34, input_context="both", hidden_size=300, n_layers=4, dropout_prob=0.3,
                           freeze_wordembed=False)
    g = pkg.PoseGenerator(a, 27, 512, 300, None, pkg.Vocab.speakers(17))
    d = pkg.ConvDiscriminator(27)
    ae = pkg.EmbeddingNet(a, 27, 34)
    gs, ds, as_ = O.make_generator_state(0), O.make_discriminator_state(1), O.make_autoencoder_state(2)
    for mod, st, n in ((g, gs, 117), (d, ds, 52), (ae, as_, 70)):
        sd = mod.state_dict()
        assert len(sd) == n and set(sd) == set(st)
        for k in sd:
            assert tuple(sd[k].shape) == tuple(st[k].shape) and sd[k].dtype == st[k].dtype, k
        mod.load_state_dict(st, strict=True)
    # the TCN's duplicated registration shares storage, like the reference
    sd = g.state_dict()
    assert sd["text_encoder.tcn.network.0.conv1.weight_v"].data_ptr() == sd["text_encoder.tcn.network.0.net.0.weight_v"].data_ptr()
    assert sum(p.numel() for p in g.parameters()) == 13_204_939 - (20000 - 512) * 300 - (1371 - 17) * 16
    assert sum(p.numel() for p in d.parameters()) == 253_950 and sum(p.numel() for p in ae.parameters()) == 190_691


def test_product_never_imports_oracle_or_reference():
    pdir = os.path.join(ROOT, "gesture-generation-from-trimodal-context_amd")
    for dirpath, _, files in os.walk(pdir):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".cpp")):
                src = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in src and "from oracle" not in src, f
                assert "sys.path" not in src or f == "__init__.py", f
                assert "/root/reference" not in src, f


def test_missing_library_fails_loudly(pkg, monkeypatch):
    monkeypatch.setattr(pkg._lib, "_lib", None)
    monkeypatch.setattr(pkg._lib, "LIB_PATH", "/nonexistent/libtrimodal_hip.so")
    try:
        pkg._lib.load()
    except RuntimeError as e:
        assert "no CPU fallback" in str(e)
    else:
        raise AssertionError("load() must raise when the HIP library is missing")


def test_every_entry_point_validates_its_arguments(pkg):
    """tests/abi_fuzz.py: every C-ABI function called with nulls / zero sizes / negative sizes / zeroed tables comes back with a status
    and never crashes -- no GPU needed, validation precedes any launch."""
    from tests.abi_fuzz import fuzz
    assert fuzz() >= 4 * len(pkg._lib.SIGNATURES)


def test_entry_points_under_host_address_sanitizer():
    """The same fuzz against the host-ASan build of the library (make -C csrc asan) in a subprocess with the ASan runtime preloaded:
    an out-of-bounds read of a launcher (problem tables, pointer arrays, split plans) would abort the subprocess with a report."""
    import glob
    import subprocess
    csrc = os.path.join(ROOT, "gesture-generation-from-trimodal-context_amd", "csrc")
    rt = sorted(glob.glob("/opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so"))
    if not rt or not os.path.exists("/opt/rocm/bin/hipcc"):
        pytest.skip("no ROCm toolchain / ASan runtime on this machine")
    subprocess.run(["make", "-C", csrc, "-j4", "asan"], check=True, capture_output=True)
    lib = os.path.join(ROOT, "gesture-generation-from-trimodal-context_amd", "libtrimodal_hip_asan.so")
    env = dict(os.environ, LD_PRELOAD=rt[-1], ASAN_OPTIONS="detect_leaks=0:abort_on_error=0:exitcode=66")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "abi_fuzz.py"), lib], env=env, capture_output=True, text=True)
    assert r.returncode == 0 and "abi fuzz ok" in r.stdout, (r.returncode, r.stdout[-500:], r.stderr[-3000:])


def test_deferred_weight_gradients_are_grouped_by_kernel_family(pkg, monkeypatch):
    """layers.tn_group_deferred (host logic, no GPU): problems that qualify for the bf16 x 3 weight-gradient kernel and those that do not are
    launched in separate groups of at most MAX_GROUP, every problem exactly once, order kept inside a kind."""
    from types import SimpleNamespace as NS
    L, ops = pkg.layers, pkg.ops
    calls = []
    monkeypatch.setattr(ops, "gemm_tn_group", lambda probs: calls.append(list(probs)))
    def prob(M, N, K, tag):
        return dict(dY=torch.zeros(M, N), A=NS(K=K, s=NS(cw=K)), dW=None, dbias=None, tag=tag)
    big = [prob(7168, 192, 128, f"big{i}") for i in range(14)]           # the discriminator's GRU layers 1..3 (ih + hh) and layer 0 hh
    small = [prob(7168, 192, 8, "ih0f"), prob(7168, 192, 8, "ih0r"), prob(7680, 8, 24, "conv2"), prob(8192, 8, 48, "conv1"), prob(512, 192, 128, "short")]
    mixed = [big[0], small[0], *big[1:8], small[1], *big[8:], *small[2:]]
    L.tn_group_deferred(mixed)
    assert [len(c) for c in calls] == [8, 6, 5]
    assert [p["tag"] for c in calls[:2] for p in c] == [f"big{i}" for i in range(14)]
    assert [p["tag"] for p in calls[2]] == ["ih0f", "ih0r", "conv2", "conv1", "short"]


def test_nt_epilogue_operands_are_checked_on_the_host(pkg, monkeypatch):
    """ops._nt_problem: res and out2 go together and every epilogue operand must be laid out exactly like the output (the wrappers accept
    CUDA tensors only; that check is lifted here to reach the layout checks without a GPU)."""
    ops, Win = pkg.ops, pkg.ops.Win
    monkeypatch.setattr(ops, "_f32", lambda t, name="tensor": None)
    x, w, out = torch.zeros(64, 32), torch.zeros(16, 32), torch.zeros(64, 16)
    q = ops._nt_problem(Win.plain(x), w, None, out, gate=torch.zeros(64, 16), res=torch.zeros(64, 16), out2=torch.zeros(64, 16), res_slope=0.0)
    assert q.gate and q.res and q.C2 and q.res_slope == 0.0
    with pytest.raises(AssertionError):
        ops._nt_problem(Win.plain(x), w, None, out, res=torch.zeros(64, 16))
    with pytest.raises(AssertionError):
        ops._nt_problem(Win.plain(x), w, None, out, gate=torch.zeros(64, 17))
    with pytest.raises(AssertionError):
        ops._nt_problem(Win.plain(x), w, None, out, gate=torch.zeros(16, 64).t())
