#!/usr/bin/env python3
"""sha256 (first 16 hex digits) over the kernel sources and their build recipe: what a profile under profiles/ was taken on.
bench.py compares it with the tree it runs from and marks counters taken on other sources as stale."""
import hashlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def kernel_sources_sha16(root=ROOT):
    h = hashlib.sha256()
    d = os.path.join(root, "zk-toolkit_amd", "csrc")
    for fn in sorted(os.listdir(d)) + ["../Makefile"]:
        p = os.path.normpath(os.path.join(d, fn))
        if os.path.isfile(p):
            h.update(fn.encode()); h.update(open(p, "rb").read())
    return h.hexdigest()[:16]


if __name__ == "__main__":
    print(kernel_sources_sha16())
