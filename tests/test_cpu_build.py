"""-m "not gpu": the build-time ISA audit of the strip kernels (scripts/audit_strip_isa.py, `make audit`).

dense_strip_impl.h keeps its bottleneck window in literal accumulator registers a160-a255 and chains asm MFMAs whose
latency hipcc does not know.  Round 3 found hipcc copying accumulators four instructions behind an asm MFMA at one geometry
(wrong rows, commit aab1e18); the audit that caught it is part of the build since round 4, and this test keeps it honest:
it must pass on the tree's own listings and FAIL on listings with the known bug patterns planted."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ISA = os.path.join(ROOT, "tennis_amd", "csrc", "isa")
UNITS = ["dense_strip_w56", "dense_strip_w28", "dense_strip_w128", "dense_strip_w64"]
AUDIT = os.path.join(ROOT, "scripts", "audit_strip_isa.py")


def _listings():
    files = [os.path.join(ISA, u + ".s") for u in UNITS]
    if not all(os.path.exists(f) for f in files):     # a tree that was built before the listings were kept
        subprocess.run(["make", "-C", ROOT, "-j8", "audit"], check=True, capture_output=True)
    return files


def test_strip_isa_audit_passes_on_the_built_objects():
    r = subprocess.run([sys.executable, AUDIT] + _listings(), capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    m = re.search(r"audited (\d+) kernels, 0 problem", r.stdout)
    assert m and int(m.group(1)) >= 30, r.stdout          # 4 map widths x 8-9 channel counts


def _one_kernel(text):
    m = re.search(r"^(\S*dense_strip_kernel\S*):", text, re.M)
    end = text.find(".end_amdhsa_kernel", m.end())
    return m.start(), text.find("\n", end) + 1


def test_strip_isa_audit_catches_planted_bugs(tmp_path):
    src = open(_listings()[1]).read()
    a, b = _one_kernel(src)
    kern = src[a:b]
    # (1) a compiler instruction (outside ;;#ASMSTART .. ;;#ASMEND) that writes a window register
    i = kern.index(";;#ASMEND")
    i = kern.index("\n", i) + 1
    bad1 = kern[:i] + "\tv_accvgpr_write_b32 a200, v0\n" + kern[i:]
    # (2) the round-3 bug: a compiler v_accvgpr_mov of an asm MFMA's result right behind the MFMA's asm block
    mm = None
    for mm in re.finditer(r";;#ASMSTART\n((?:(?!;;#ASM).*\n)*?)\t?;;#ASMEND\n", kern):
        dst = re.search(r"v_mfma_f32_32x32x16_f16 a\[(\d+):\d+\]", mm.group(1))
        if dst and "s_nop 15" not in mm.group(1):
            break
    assert mm is not None and dst
    bad2 = kern[:mm.end()] + "\tv_accvgpr_mov_b32 a0, a%s\n" % dst.group(1) + kern[mm.end():]
    # (3) a spill
    bad3 = re.sub(r"(\.amdhsa_private_segment_fixed_size\s+)0", r"\g<1>64", kern)
    assert bad3 != kern
    for name, text in (("window", bad1), ("early_read", bad2), ("spill", bad3)):
        f = tmp_path / (name + ".s")
        f.write_text(text)
        r = subprocess.run([sys.executable, AUDIT, str(f)], capture_output=True, text=True)
        assert r.returncode == 1 and "problem" in r.stdout and "0 problem" not in r.stdout, (name, r.stdout[-400:])
    # and a file without any strip kernel is an error, not a pass
    f = tmp_path / "empty.s"
    f.write_text("\ts_endpgm\n")
    assert subprocess.run([sys.executable, AUDIT, str(f)], capture_output=True).returncode == 1
