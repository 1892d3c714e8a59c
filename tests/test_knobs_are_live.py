"""Every FZ_* / FUSION_HIP_* environment name that tests/, tools/ or bench.py SET is one the product really reads (VERDICT r04 #5:
round 4 removed FZ_KEYGEN_BCAST_GENERAL and FZ_SAMPLER_ONE_KERNEL from the library and left tests parametrised over them --
both parameter values ran the same kernel and six "passes" asserted nothing).  CPU only: the sources are read as text."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NAME = r"(?:FZ|FUSION_HIP)_[A-Z][A-Z0-9_]*"


def _read(*parts):
    with open(os.path.join(ROOT, *parts)) as fh:
        return fh.read()


def names_the_product_reads():
    """getenv("...") / knob("...") in csrc/, os.environ.get / os.environ[...] / getenv in the Python packages and bench.py"""
    found = set()
    csrc = os.path.join(ROOT, "fusion-cryptography_amd", "csrc")
    for f in os.listdir(csrc):
        if f.endswith((".hip", ".cpp", ".h")):
            found |= set(re.findall(r'(?:getenv|knob)\(\s*"(' + NAME + r')"', _read("fusion-cryptography_amd", "csrc", f)))
    py = [os.path.join(b, f) for b, _, fs in os.walk(os.path.join(ROOT, "fusion-cryptography_amd")) for f in fs if f.endswith(".py")]
    py += [os.path.join(ROOT, "bench.py"), os.path.join(ROOT, "tools", "bench_legs.py")]
    for path in py:
        text = open(path).read()
        found |= set(re.findall(r'environ(?:\.get|\.setdefault)?\(\s*"(' + NAME + r')"', text))
        found |= set(re.findall(r'environ\[\s*"(' + NAME + r')"\s*\]', text))
        found |= set(re.findall(r'getenv\(\s*"(' + NAME + r')"', text))
    return found


def names_the_tests_set():
    """string literals that look like a knob in tests/, tools/*.py, tools/*.sh and the docs' knob table"""
    used = {}
    for sub in ("tests", "tools"):
        for b, _, fs in os.walk(os.path.join(ROOT, sub)):
            if "__pycache__" in b:
                continue
            for f in fs:
                if not f.endswith((".py", ".sh")) or f == os.path.basename(__file__):
                    continue
                text = open(os.path.join(b, f)).read()
                # comments and docstrings may MENTION a removed knob (history); what counts is a quoted name or a VAR=value prefix
                for m in re.finditer(r'["\'](' + NAME + r')["\']|\b(' + NAME + r')=[0-9A-Za-z$"]', text):
                    used.setdefault(m.group(1) or m.group(2), set()).add(os.path.join(sub, f))
    return used


def test_every_knob_a_test_or_tool_sets_is_read_by_the_product():
    live = names_the_product_reads()
    assert {"FZ_NTT_KERNEL", "FZ_NTT_ROWS", "FZ_UNFUSED", "FZ_POOL_MB", "FZ_KECCAK", "FZ_HIP_RUNTIME", "FUSION_HIP_LIB"} <= live, sorted(live)
    # names that are not knobs of the product: C macros / status codes / queue flags, and tests' own switches
    not_knobs = {n for n in names_the_tests_set() if n.startswith(("FZ_E_", "FZ_OK", "FZ_VERDICT_", "FZ_QUEUE_", "FZ_API", "FZ_OP_", "FZ_TEST_"))}
    dead = {n: sorted(fs) for n, fs in names_the_tests_set().items() if n not in live and n not in not_knobs}
    assert not dead, f"set by tests / tools but read by nothing in the product: {dead}"


def test_design_lists_exactly_the_library_knobs():
    """DESIGN.md section 10's table = what fz_ctx_create / the loaders read (bench.py's FZ_BENCH_* switches are listed below it)"""
    design = _read("DESIGN.md")
    sec = design[design.index("## 10. Knobs"):]
    sec = sec[:sec.index("\n## ", 5)] if "\n## " in sec[5:] else sec
    table = set(re.findall(r"^\| `(" + NAME + r")\b", sec, re.M))
    lib = {n for n in names_the_product_reads() if not n.startswith("FZ_BENCH_")}
    assert table == lib, (sorted(table - lib), sorted(lib - table))
