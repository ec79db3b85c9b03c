"""Invariants of hand-counted code that nothing else checks: k_syrk_bf16x6's `s_waitcnt vmcnt(63 / 22 / 6)`
assume exactly SIX LDS-DMA loads per wave and chunk, and that every other vector-memory access of the kernel is a
global_* one (counted in order by vmcnt; a flat_* access returns out of order and a compiler-added one would break the
count).  The built library's own gfx950 code object is disassembled and checked.  CPU only."""
import os
import re
import struct
import subprocess
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "ekf-monoslam_for_3d-reconstruction_amd", "lib", "libekfslam_hip.so")
OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"


def _code_object():
    blob = open(LIB, "rb").read()
    i = blob.find(b"__CLANG_OFFLOAD_BUNDLE__")
    assert i >= 0, "no offload bundle in the library"
    n = struct.unpack_from("<Q", blob, i + 24)[0]
    off = i + 32
    for _ in range(n):
        o, s, ln = struct.unpack_from("<QQQ", blob, off)
        off += 24
        name = blob[off:off + ln]
        off += ln
        if b"gfx950" in name:
            return blob[i + o:i + o + s]
    raise AssertionError("no gfx950 code object in the bundle")


@pytest.fixture(scope="module")
def disassembly():
    if not (os.path.exists(LIB) and os.path.exists(OBJDUMP)):
        pytest.skip("library or llvm-objdump missing")
    with tempfile.NamedTemporaryFile(suffix=".co") as f:
        f.write(_code_object())
        f.flush()
        out = subprocess.run([OBJDUMP, "-d", "--no-show-raw-insn", f.name], capture_output=True, text=True, check=True).stdout
    funcs = {}
    cur = None
    for line in out.splitlines():
        m = re.match(r"^[0-9a-f]+ <(.+)>:$", line)
        if m:
            cur = m.group(1)
            funcs[cur] = []
        elif cur and line.strip():
            funcs[cur].append(line.strip())
    return funcs


def _kernel(funcs, needle):
    names = [n for n in funcs if needle in n]
    assert names, f"{needle} not found in the code object"
    return names


def test_syrk6_only_counts_what_it_issues(disassembly):
    for name in _kernel(disassembly, "k_syrk_bf16x6ILi0"):
        body = disassembly[name]
        flat = [l for l in body if re.search(r"\bflat_(load|store|atomic)", l)]
        assert not flat, f"{name}: flat_* accesses return out of order and break the counted waits: {flat[:3]}"
        scratch = [l for l in body if "scratch_" in l]
        assert not scratch, f"{name}: spills would add uncounted vector-memory accesses: {scratch[:3]}"
        dma = [l for l in body if re.search(r"global_load_lds_dwordx4|global_load_dwordx4 .* lds", l)]
        # issue() is inlined at four places (two in front of the first tile, two inside the K loop): six pieces each
        assert len(dma) > 0 and len(dma) % 6 == 0, f"{name}: {len(dma)} LDS-DMA loads, not a multiple of six"
        waits = [l for l in body if re.search(r"s_waitcnt vmcnt\((63|22|6)\)", l)]
        assert len(waits) >= 3, f"{name}: the counted waits are gone ({len(waits)})"


def test_chain_kernel_hand_overs_are_write_through(disassembly):
    """Every access of Y / Dinv inside the persistent chain kernel is an sc1 buffer access (the hand-over form of
    MI355X_MICROARCH.md needs EVERY load of handed-over bytes to bypass L1 and every store to be write-through)."""
    # (chain::trail<0> -- ordinary accesses -- is k_trail_diag's instantiation: its hand-overs are launch boundaries)
    names = [n for n in disassembly if "chain" in n and ("trailILi16E" in n or "panel" in n or "crit_" in n)]
    assert names and any("trailILi16E" in n for n in names)
    for name in names:
        for l in disassembly[name]:
            if re.search(r"\bbuffer_(load|store)_dword", l):
                assert " sc1" in l, f"{name}: {l}"
            assert not re.search(r"\bflat_load", l), f"{name}: {l}"
