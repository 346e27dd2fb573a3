"""Hash of kernel sources as the compiler sees them: comments and blank lines do not count, so that the counter records of
profiles/pmc_traffic.json (measured on a given kernel) survive a comment edit but not a code edit.  Used by bench.py and
tools/make_pmc_records.py, which must agree."""
import hashlib
import os
import re

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
_BLOCK = re.compile(r"/\*.*?\*/", re.S)
_LINE = re.compile(r"//[^\n]*")


def code_only(text):
    text = _LINE.sub("", _BLOCK.sub("", text))   # (the kernel sources hold no string literal with a comment marker in it)
    return "\n".join(l.rstrip() for l in text.splitlines() if l.strip())


def kernel_source_sha16(files):
    h = hashlib.sha256()
    for f in files:
        h.update(code_only(open(os.path.join(CSRC, f), encoding="utf-8").read()).encode())
    return h.hexdigest()[:16]
