"""Which kernel sources a committed measurement belongs to: sha256 over the files that define the device code
(flash-attention-cuda-c_amd/csrc/*, helpers.hpp, include/flash_attention.h), in sorted order.  profiles/summarize.py stores
it next to the HBM-traffic figure; bench.py reports that figure as `roofline.traffic` only when the hash still matches
the sources the library is built from."""
import glob
import hashlib
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def kernel_source_files():
    pkg = os.path.join(ROOT, "flash-attention-cuda-c_amd")
    files = sorted(glob.glob(os.path.join(pkg, "csrc", "*")))
    files += [os.path.join(pkg, "helpers.hpp"), os.path.join(ROOT, "include", "flash_attention.h")]
    return [f for f in files if os.path.isfile(f)]


def csrc_sha256():
    h = hashlib.sha256()
    for f in kernel_source_files():
        h.update(os.path.relpath(f, ROOT).encode())
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()


if __name__ == "__main__":
    print(csrc_sha256())
