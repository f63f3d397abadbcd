"""Which kernel sources a committed measurement belongs to: sha256 over the CODE of the files that define the device code
(flash-attention-cuda-c_amd/csrc/*, helpers.hpp, include/flash_attention.h), in sorted order -- comments and white space are
stripped first, so that rewording a comment does not orphan a measurement while any change to the code still does.  profiles/summarize.py stores
it next to the HBM-traffic figure; bench.py reports that figure as `roofline.traffic` only when the hash still matches
the sources the library is built from."""
import glob
import hashlib
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def kernel_source_files():
    pkg = os.path.join(ROOT, "flash-attention-cuda-c_amd")
    files = sorted(glob.glob(os.path.join(pkg, "csrc", "*")))
    files += [os.path.join(pkg, "helpers.hpp"), os.path.join(ROOT, "include", "flash_attention.h")]
    return [f for f in files if os.path.isfile(f)]


_TOKEN = re.compile(r'''"(?:\\.|[^"\\])*"|'(?:\\.|[^'\\])*'|//[^\n]*|/\*.*?\*/''', re.S)


def code_only(text):
    """The text without C / C++ comments (string and character literals are kept as they are) and without white space."""
    text = _TOKEN.sub(lambda m: m.group(0) if m.group(0)[0] in "\"'" else " ", text)
    return re.sub(r"\s+", "", text)


def csrc_sha256():
    h = hashlib.sha256()
    for f in kernel_source_files():
        h.update(os.path.relpath(f, ROOT).encode())
        with open(f, "r", encoding="utf-8") as fh:
            h.update(code_only(fh.read()).encode())
    return h.hexdigest()


if __name__ == "__main__":
    print(csrc_sha256())
