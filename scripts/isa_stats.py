"""Instruction statistics of one kernel in a hipcc -save-temps .s file (no GPU needed).

  python scripts/isa_stats.py <file.s> <substring of the kernel symbol> [--dump] [--loop]

Prints the instruction histogram; --dump writes the kernel's instruction stream (comments stripped) to stdout;
--loop restricts both to the largest loop body (label .. backward branch)."""
import collections
import re
import sys


def kernel_body(path, key):
    s = open(path).read()
    m = None
    for mm in re.finditer(r"^(\S*%s\S*):" % re.escape(key), s, re.M):
        m = mm
        break
    if m is None:
        raise SystemExit("no symbol containing %r" % key)
    end = s.find(".end_amdhsa_kernel", m.end())
    body = s[m.end():s.rfind("s_endpgm", m.end(), end) + 8]
    return m.group(1), body


def instructions(body):
    out = []
    for l in body.split("\n"):
        l = l.split(";")[0].strip()
        if not l or l.startswith("."):
            if l.startswith(".LBB") and l.endswith(":"):
                out.append(l)
            continue
        out.append(l)
    return out


def main():
    path, key = sys.argv[1], sys.argv[2]
    name, body = kernel_body(path, key)
    ins = instructions(body)
    if "--loop" in sys.argv:
        labels = {l[:-1]: i for i, l in enumerate(ins) if l.endswith(":")}
        best = None
        for i, l in enumerate(ins):
            m = re.match(r"s_cbranch_\w+\s+(\S+)", l)
            if m and m.group(1) in labels and labels[m.group(1)] < i:
                span = (labels[m.group(1)], i)
                if best is None or span[1] - span[0] > best[1] - best[0]:
                    best = span
        if best:
            ins = ins[best[0]:best[1] + 1]
    if "--dump" in sys.argv:
        print("\n".join(ins))
        return
    c = collections.Counter(l.split()[0] for l in ins if not l.endswith(":"))
    print(name, "instructions:", sum(c.values()))
    for k, v in c.most_common(60):
        print("  %-40s %d" % (k, v))


if __name__ == "__main__":
    main()
