"""Identity of the code a counter summary was measured on: sha256 over EVERY file the kernel's translation unit includes
(the quoted #include closure of its .hip file inside simpleworks_amd/csrc and include/, asm .inc texts included).
tools/pmc_summaries.py stores it with each summary; bench.py recomputes it and reports `..._stale` when it differs."""
import hashlib
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "simpleworks_amd", "csrc")
INC = os.path.join(ROOT, "include")
# translation unit of each measured kernel (launch geometry and kernels live in the same file)
UNITS = {"msm_accumulate": "msm.hip", "ntt_pass": "ntt.hip", "spmv": "spmv.hip"}


def include_closure(unit):
    """sorted list of paths (relative to the repo root) the unit includes, itself first"""
    seen, todo = [], [os.path.join(CSRC, unit)]
    while todo:
        p = todo.pop()
        if p in seen or not os.path.isfile(p):
            continue
        seen.append(p)
        for m in re.finditer(r'^\s*#\s*include\s+"([^"]+)"', open(p, errors="replace").read(), re.M):
            for base in (os.path.dirname(p), CSRC, INC):
                q = os.path.normpath(os.path.join(base, m.group(1)))
                if os.path.isfile(q):
                    todo.append(q)
                    break
    if not seen:  # the kernel's translation unit is not there (an installed tree without csrc/)
        return None
    first = seen[0]
    return [os.path.relpath(first, ROOT)] + sorted(os.path.relpath(p, ROOT) for p in seen[1:])


def unit_sha16(kernel):
    h = hashlib.sha256()
    try:
        closure = include_closure(UNITS[kernel])
        if closure is None:
            return None
        for rel in closure:
            h.update(rel.encode() + b"\0")
            h.update(open(os.path.join(ROOT, rel), "rb").read())
    except (OSError, KeyError, IndexError):
        return None
    return h.hexdigest()[:16]


if __name__ == "__main__":
    for k in UNITS:
        print(k, unit_sha16(k), include_closure(UNITS[k]))
