"""Build-time ISA assertions (cross-compiled, no GPU): properties of the emitted gfx950 code that the C++ source cannot
guarantee by itself and that a silent compiler change would break."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "autostyle-tts_amd", "csrc")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


def _asm(src, tmp_path):
    if not (shutil.which(HIPCC) or os.path.exists(HIPCC)):
        pytest.skip("hipcc not available")
    out = tmp_path / (os.path.basename(src) + ".s")
    subprocess.run([HIPCC, "-O3", "-std=c++17", "--offload-arch=gfx950", "-mllvm", "-amdgpu-mfma-vgpr-form=1", "-mllvm",
                    "-amdgpu-kernarg-preload-count=16", "-S", "--cuda-device-only", src, "-o", str(out)],
                   check=True, cwd=CSRC, timeout=600)
    return out.read_text()


def _kernels(asm, pattern):
    """{mangled name: body text} of the kernels whose name matches pattern."""
    out = {}
    for m in re.finditer(r"^(_Z\w+):\s*;\s*@\1\n(.*?)^\s*\.end_amdhsa_kernel", asm, flags=re.S | re.M):
        if re.search(pattern, m.group(1)):
            out[m.group(1)] = m.group(2)
    return out


def test_splitk_partials_are_drained_before_the_arrival_barrier(tmp_path):
    """Split-K hand-off of gemm_skinny16 (ops_gemm.hip): every wave's sc1 partial-sum store must be acknowledged
    (s_waitcnt vmcnt(0)) BEFORE the workgroup barrier that precedes the arrival-counter bump -- s_barrier does not wait
    for vmcnt and the compiler emits no wait of its own there (store -> s_barrier -> atomic was racy)."""
    ks = _kernels(_asm(os.path.join(CSRC, "ops_gemm.hip"), tmp_path), r"gemm_skinny16")
    assert len(ks) == 4, sorted(ks)
    for name, body in ks.items():
        lines = [l.strip() for l in body.splitlines()]
        stores = [i for i, l in enumerate(lines) if l.startswith("global_store_dword") and " sc1" in l]
        atomics = [i for i, l in enumerate(lines) if l.startswith("global_atomic_add")]
        assert stores and atomics, name
        # layout order == program order here: the partial store's block falls through to the barrier and the atomic
        st = max(i for i in stores if i < atomics[0])
        between = lines[st + 1:atomics[0]]
        assert "s_barrier" in between, name
        bar = between.index("s_barrier")
        assert any(l.startswith("s_waitcnt vmcnt(0)") for l in between[:bar]), f"{name}: no vmcnt(0) drain between the sc1 store and s_barrier"
        # and the last arriver takes an agent-scope acquire before re-reading the partials
        assert any(l.startswith("buffer_inv sc1") for l in lines[atomics[0]:]), name


def test_decode_step_kernels_get_their_leading_arguments_preloaded(tmp_path):
    """lm_gemv / lm_attn (lm_step.hip) pass what their first global loads need as explicit leading parameters so that the command
    processor preloads them into SGPRs (-amdgpu-kernarg-preload-count, csrc/Makefile): 13 / 14 dwords (14 is the hardware's limit).  A by-value struct
    parameter is NOT preloaded (length 0), so a refactor back to `lm_gemv(GemvArgs)` would silently lose 5 % of the decode step."""
    with open(os.path.join(CSRC, "Makefile")) as f:
        assert "-amdgpu-kernarg-preload-count=16" in f.read()
    asm = _asm(os.path.join(CSRC, "lm_step.hip"), tmp_path)
    found = {}
    for m in re.finditer(r"\.amdhsa_kernel (_Z\w+)\n(.*?)\.end_amdhsa_kernel", asm, flags=re.S):
        n = re.search(r"\.amdhsa_user_sgpr_kernarg_preload_length\s+(\d+)", m.group(2))
        found[m.group(1)] = int(n.group(1)) if n else 0
    gemv = {k: v for k, v in found.items() if "lm_gemv" in k}
    attn = {k: v for k, v in found.items() if "lm_attn" in k}
    assert len(gemv) >= 10 and all(v == 13 for v in gemv.values()), gemv
    assert len(attn) == 1 and all(v == 14 for v in attn.values()), attn
