"""ISA-level checks of the cross-workgroup protocols (CPU only: hipcc cross-compiles gfx950 without a GPU).

The fused verification kernel combines its workgroups through agent-scope atomics with RELAXED ordering (a device-scope
fence is an L2 write-back on this multi-XCD part; ADVICE round 1 called the resulting reliance on hardware behaviour
fragile).  What the correctness argument needs from the generated code is pinned here, so that a compiler change that
breaks it fails the CPU suite instead of producing a rare wrong verdict:
  * the adds into the shared accumulator are RETURNING atomics (`sc0`: the old value comes back = the add was performed),
  * between the last of them and the workgroup barrier the wave executes `s_waitcnt vmcnt(0)`,
  * the arrival counter is bumped only after that barrier,
  * the default build contains no `buffer_wbl2` / `buffer_inv` (the fence-free form is what is being measured), the
    FZ_VERIFY_ORDERED instantiation contains both."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "fusion-cryptography_amd", "csrc")


@pytest.fixture(scope="module")
def asm(tmp_path_factory):
    out = tmp_path_factory.mktemp("isa") / "fz_ntt.s"
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc")
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-S", "--cuda-device-only",
                           os.path.join(CSRC, "fz_ntt.hip"), "-o", str(out)], stderr=subprocess.DEVNULL)
    return open(out).read()


def bodies(asm, needle):
    """{mangled name: [instructions]} of every kernel whose name contains `needle`"""
    out = {}
    for m in re.finditer(r"^(_ZN\S*" + needle + r"\S*):\s*;.*?$(.*?)s_endpgm", asm, re.S | re.M):
        ins = [ln.strip() for ln in m.group(2).splitlines() if ln.startswith("\t") and not ln.strip().startswith((".", ";"))]
        out[m.group(1)] = ins
    return out


def test_verify_fused_orders_its_adds_before_the_arrival(asm):
    ks = bodies(asm, "verify_fused")
    # degree 64 / 256 x general / pseudo-Mersenne multiply x int32 / int64 rows x ordered / not x integer / fp64 accumulation of
    # A * sigma (round 3 also carried one / two row groups per iteration x twiddles as pairs / w alone: 128 instantiations)
    assert len(ks) == 32, sorted(ks)
    for name, ins in ks.items():
        ordered = re.search(r"verify_fusedILi\dELb\dE[il]Lb(\d)ELb\dE", name).group(1) == "1"     # fourth template argument
        adds = [i for i, s in enumerate(ins) if s.startswith("global_atomic_add_f64")]
        assert adds, name
        assert all(" sc0" in ins[i] for i in adds), (name, [ins[i] for i in adds])      # returning form
        after = ins[adds[-1] + 1:]
        bar = next(i for i, s in enumerate(after) if s.startswith("s_barrier"))
        assert any(s.startswith("s_waitcnt vmcnt(0)") for s in after[:bar]), (name, after[:bar])
        arrive = [i for i, s in enumerate(ins) if s.startswith("global_atomic_add ") or s.startswith("global_atomic_add_u32")]
        assert arrive and arrive[0] > adds[-1] + 1 + bar, name
        fences = [s for s in ins if s.startswith(("buffer_wbl2", "buffer_inv"))]
        assert bool(fences) == ordered, (name, fences)


def test_aggregate_onepass_instantiations_do_not_spill_and_add_with_returning_atomics(tmp_path_factory):
    """aggregate_onepass<8, RAG, SIGN, AR>: the nine instantiations the launcher can pick (rows per column block 4 / 3 / 2 x plain /
    ragged / fused signing) keep their state in registers -- no scratch (a spill in the signer loop would double the launch) --
    and every cross-workgroup sum is added with a RETURNING 64-bit atomic (the arrival count travels in the word the add returns),
    paired with the one non-returning add that re-arms the word"""
    out = tmp_path_factory.mktemp("isa") / "fz_pointwise.s"
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc")
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-S", "--cuda-device-only",
                           os.path.join(CSRC, "fz_pointwise.hip"), "-o", str(out)], stderr=subprocess.DEVNULL)
    text = open(out).read()
    meta = {}
    for m in re.finditer(r"\.name:\s+(\S*aggregate_onepass\S*)\n(.*?)\.wavefront_size", text, re.S):
        f = dict(re.findall(r"\.(vgpr_count|vgpr_spill_count|sgpr_spill_count|private_segment_fixed_size):\s+(\d+)", m.group(2)))
        meta[m.group(1)] = {k: int(v) for k, v in f.items()}
    assert len(meta) == 9, sorted(meta)
    for name, f in meta.items():
        assert f["vgpr_spill_count"] == 0 and f["sgpr_spill_count"] == 0 and f["private_segment_fixed_size"] == 0, (name, f)
        assert f["vgpr_count"] <= 256, (name, f)                 # two waves per SIMD: one 8-wave workgroup per CU
    ks = bodies(text, "aggregate_onepass")
    assert len(ks) == 9
    for name, ins in ks.items():
        adds = [s for s in ins if s.startswith("global_atomic_add_x2")]
        returning = [s for s in adds if " sc0" in s]           # the sum's add: its return value says who arrived last
        rearm = [s for s in adds if " sc0" not in s]           # the last arrival's subtraction that re-arms the word: result unused
        assert returning and len(returning) == len(rearm), (name, adds)
