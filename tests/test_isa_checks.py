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
    processor preloads them into SGPRs (-amdgpu-kernarg-preload-count, csrc/Makefile): 14 dwords each, the hardware's limit.  A by-value struct
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
    assert len(gemv) >= 10 and all(v == 14 for v in gemv.values()), gemv
    assert len(attn) == 1 and all(v == 14 for v in attn.values()), attn


def _longest_load_batch(body):
    """Longest run of global loads in the kernel's text that no FULL wait interrupts: s_waitcnt vmcnt(0) or any s_waitcnt lgkmcnt (a scalar
    wait covers every scalar load in flight, i.e. a cold kernarg miss).  Counted vector waits (vmcnt(N > 0)) leave loads in flight and do
    not end a run.  The kernarg-preload compatibility prologue in front of the first numbered block is skipped."""
    lines = body.splitlines()
    start = next((i for i, l in enumerate(lines) if re.match(r"\.LBB\d+_0:", l.strip())), 0)
    best = run = 0
    for l in lines[start:]:
        t = l.strip()
        if t.startswith("global_load"):
            run += 1
            best = max(best, run)
        elif t.startswith("s_waitcnt") and ("lgkmcnt" in t or "vmcnt(0)" in t):
            run = 0
    return best


def test_decode_step_kernels_issue_their_loads_in_one_batch(tmp_path):
    """What round 4's ISA audit established (EXPERIMENTS.md G), as an assertion on the emitted code of the lm_gemv variants of a decode step
    and of lm_attn: the up-front loads -- input rows / pieces AND the weight lines -- form one batch that no full wait interrupts.  Each of
    these once broke it: loads inside bounds checks (one dependent round trip per block), a guarded load block merged with its guarded
    consumer, the wait for a gathered row index between the rows of a wave, a struct field missing from the argument pin (its SGPR reused
    under the in-flight scalar load), an implicit argument (gridDim) read in front of the loads."""
    asm = _asm(os.path.join(CSRC, "lm_step.hip"), tmp_path)
    want = {
        r"lm_gemvILi1ELi0ELi0ELi2ELi1E": 8 + 4,      # QKV / FFN-in / head at <= 16 rows: 2 rows x 4 float4 + 2 weight lines x 2 fragments
        r"lm_gemvILi2ELi0ELi0ELi2ELi1E": 16 + 4,     # the same at 32 rows: the 4 rows of a wave
        r"lm_gemvILi1ELi1ELi2ELi2ELi1E": 10 + 4,     # out-projection at <= 8 rows: 2 pieces x 5 partial loads + weights
        r"lm_gemvILi1ELi1ELi1ELi2ELi1E": 8 + 4,      # FFN-out (one K slice) at <= 8 rows: 8 fp16 pieces + weights
        r"lm_attn": 12,                              # one chunk of K / position / V rows (the query loads sit in front of an optional path)
    }
    for pat, batch in want.items():
        ks = _kernels(asm, pat)
        assert len(ks) == 1, (pat, sorted(ks))
        name, body = next(iter(ks.items()))
        assert _longest_load_batch(body) >= batch, (name, _longest_load_batch(body), batch)


def test_render_kernels_issue_their_prologue_loads_in_one_batch(tmp_path):
    """The same property for the flow / vocoder kernels whose prologues were chains of guarded loads (EXPERIMENTS.md G): the frame rows of
    rconv_lds (+ its two units of weights) and of conv_lds' staging pass, the weight fragments + first row chunks of tfm_attn_fused, the A / B
    chunks of a gemm_tile K tile."""
    cases = [("ops_resnet_conv.hip", r"rconv_ldsILi256ELb0E", 5 + 32), ("ops_resnet_conv.hip", r"rconv_ldsILi512ELb0E", 9 + 32),
             ("ops_resnet_conv.hip", r"rconv_ldsILi256ELb1E", 5 + 16),      # (two workgroups per CU: weights in half units)
             ("ops_conv_lds.hip", r"conv_ldsILi128ELi128E", 12), ("ops_tfm_fused.hip", r"tfm_attn_fusedILb0E", 16 + 4),
             ("ops_tfm_fused.hip", r"tfm_attn_fusedILb1E", 16 + 4),
             ("ops_gemm.hip", r"gemm_tileILi2ELi2ELi1ELi1ELb0ELi128E", 8 + 4)]
    cache = {}
    for src, pat, batch in cases:
        if src not in cache:
            cache[src] = _asm(os.path.join(CSRC, src), tmp_path)
        ks = _kernels(cache[src], pat)
        assert len(ks) == 1, (pat, sorted(ks))
        name, body = next(iter(ks.items()))
        assert _longest_load_batch(body) >= batch, (name, _longest_load_batch(body), batch)


def test_no_volatile_system_scope_loads_in_the_hot_kernels(tmp_path):
    """A `volatile` global load compiles to a system-scope `flat_load ... sc0 sc1` followed at once by s_waitcnt vmcnt(0) lgkmcnt(0): the L2
    prefetches of the flow kernels stalled their issuing waves on an HBM miss that way.  They are untracked asm loads now (xlane.h)."""
    for src in ("ops_tfm_fused.hip", "ops_resnet_conv.hip", "lm_step.hip", "ops_conv_lds.hip"):
        asm = _asm(os.path.join(CSRC, src), tmp_path)
        assert not re.search(r"(flat|global)_load_\w+ [^\n]*sc0 sc1", asm), src


def test_untracked_prefetch_loads_keep_their_destination_register_to_themselves(tmp_path):
    """csrc/xlane.h prefetch_line(): an L2 prefetch issued from inline asm (`global_load_dword vN, ...` between ;;#ASMSTART / ;;#ASMEND)
    that the compiler's vmcnt bookkeeping cannot see -- nothing ever waits for it, and its write-back lands whenever the memory system
    answers.  That is only safe while vN holds nothing else from the load to the end of the kernel: prefetch_keep() asks the register
    allocator for exactly that, but a copy, split or spill under pressure would hand vN to another value that the late write-back then
    corrupts, with every functional test still green on most runs.  Held here on the emitted code of every kernel that uses
    it: NO instruction reachable from the asm load (control-flow graph over fall-through and branch targets, so loops and out-of-line
    blocks count) names vN, alone or inside a register range, and the kernel uses no scratch (a spill could park vN's owner there)."""
    total = 0
    for src, pattern in (("ops_tfm_fused.hip", r"tfm_attn_fused|tfm_ffn_fused"), ("ops_resnet_conv.hip", r"rconv_lds")):
        ks = _kernels(_asm(os.path.join(CSRC, src), tmp_path), pattern)
        assert ks, src
        for name, body in ks.items():
            raw = body.splitlines()
            code = [l.split(";")[0].strip() for l in raw]
            labels = {m.group(1): i for i, l in enumerate(code) for m in [re.match(r"^(\.LBB\w+):", l)] if m}

            def successors(i):
                ins = code[i]
                if ins.startswith("s_endpgm"):
                    return []
                b = re.match(r"s_(c?)branch\w*\s+(\.LBB\w+)", ins)
                out = []
                if b:
                    out.append(labels[b.group(2)])
                    if not b.group(1):          # s_branch: unconditional, no fall-through
                        return out
                if i + 1 < len(code):
                    out.append(i + 1)
                return out

            for i, l in enumerate(raw):
                if "ASMSTART" not in l or i + 1 >= len(raw):
                    continue
                m = re.match(r"\s*global_load_dword v(\d+),", raw[i + 1])
                if not m:
                    continue
                total += 1
                r = int(m.group(1))
                seen, todo = set(), [i + 2]
                while todo:
                    j = todo.pop()
                    if j in seen or j >= len(code):
                        continue
                    seen.add(j)
                    todo.extend(successors(j))
                assert any(code[j].startswith("s_endpgm") for j in seen), name
                for j in sorted(seen):
                    if j == i + 1:
                        continue                # (the load itself, when a loop leads back to it: it only rewrites its own register)
                    ins = code[j]
                    named = any(int(a.group(1)) == r for a in re.finditer(r"\bv(\d+)\b", ins)) or \
                        any(int(a.group(1)) <= r <= int(a.group(2)) for a in re.finditer(r"\bv\[(\d+):(\d+)\]", ins))
                    assert not named, f"{name}: v{r} (destination of the untracked prefetch at line {i + 1}) is used again at line {j}: {ins}"
            assert "scratch_" not in body, f"{name}: uses scratch"
    assert total >= 6, total          # 3 in tfm_attn_fused, 1 in each tfm_ffn_fused variant, 2 per rconv_lds variant


def test_eight_phase_ring_steady_loop_has_no_slack_eaters(tmp_path):
    """gemm_ring8 (ops_gemm.hip): the hand-off between its two wave groups has no slack -- a phase's last MFMA issues 32 cycles before the
    pipe drains, so every instruction between it and s_barrier idles the MFMA pipe (the first cut, with a counted-wait ladder and a
    divergent group branch in every phase, ran 3-5 % BEHIND the one-barrier ring; the steady-state loop with constant waits runs 8-10 %
    ahead: EXPERIMENTS.md P).  Hold the emitted steady-state K-tile loop to that shape: 32 MFMAs in four clusters of eight, eight raw
    barriers, 24 fragment reads, 8 LDS-DMA requests, waits with IMMEDIATE counts only (vmcnt(8) / vmcnt(10), never 0), scalar group
    branches (no exec-mask save), no scratch traffic, and nothing but s_setprio / a scalar branch / the counted wait between a cluster's
    last MFMA and the barrier behind it (at most a dozen instructions: the first cut had 25)."""
    ks = _kernels(_asm(os.path.join(CSRC, "ops_gemm.hip"), tmp_path), r"gemm_ring8")
    assert len(ks) == 1, sorted(ks)
    lines = [l.split(";")[0].strip() for l in next(iter(ks.values())).splitlines()]
    lines = [l for l in lines if l]
    # the steady-state loop: from a loop-header label to its backward branch, containing exactly 32 MFMAs
    labels = {l[:-1]: i for i, l in enumerate(lines) if re.match(r"\.LBB\d+_\d+:$", l)}
    loops = []
    for i, l in enumerate(lines):
        m = re.match(r"s_cbranch_\w+ (\.LBB\d+_\d+)$", l)
        if m and m.group(1) in labels and labels[m.group(1)] < i:
            body = lines[labels[m.group(1)]:i + 1]
            if sum(x.startswith("v_mfma") for x in body) == 32:
                loops.append(body)
    assert loops, "no K-tile loop with 32 MFMAs found"
    steady = min(loops, key=len)          # (the tail loop carries the counted-wait ladder and is longer)
    count = lambda pred: sum(bool(pred(x)) for x in steady)
    assert count(lambda x: x == "s_barrier") == 8
    assert count(lambda x: x.startswith("ds_read_b128")) == 24
    assert count(lambda x: x.startswith("global_load_lds_dwordx4")) == 8
    assert count(lambda x: x.startswith("scratch_")) == 0
    assert count(lambda x: "saveexec" in x) == 0, "the group branch must be scalar (wid through readfirstlane)"
    waits = [x for x in steady if x.startswith("s_waitcnt vmcnt")]
    assert waits and all(re.fullmatch(r"s_waitcnt vmcnt\((8|10)\)", x) for x in waits), waits
    # between the last MFMA of a cluster and the next barrier
    n = len(steady)
    for i, x in enumerate(steady):
        if x.startswith("v_mfma") and not steady[(i + 1) % n].startswith("v_mfma"):
            j = i + 1
            while steady[j % n] != "s_barrier":          # (the loop is cyclic: its last barrier may sit in front of the header's reads)
                # (the compiler may lift one address add of the next phase's DMA request above the barrier)
                assert re.match(r"(s_setprio|s_cbranch_|s_waitcnt vmcnt|s_and_b64 vcc|s_andn2_b64 vcc|\.LBB|v_cndmask|v_cmp_ne_u32|s_mov_b64|"
                                r"v_lshl_add_u64|s_add_i32|s_add_u32|s_addc_u32|v_add_u32|s_cmp_)", steady[j % n]), f"'{steady[j % n]}' between a phase's last MFMA and its barrier"
                j += 1
            assert j - i <= 12, (steady + steady)[i:j + 1]
