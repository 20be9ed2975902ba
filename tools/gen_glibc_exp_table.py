#!/usr/bin/env python3
"""The 2^(k/128) table of the exp() that glibc >= 2.28 ships (its `__exp_data.tab`), from its defining formula.

    python tools/gen_glibc_exp_table.py > picasso_amd/csrc/libm_glibc_exp_table.inc
    python tools/gen_glibc_exp_table.py --check /lib/x86_64-linux-gnu/libm.so.6      # the same 2 KiB are inside that file

For k = 0 .. 127, with e = 2^(k/128): H = RN(e), T = RN((e - H) / H); the table holds the bits of T and the bits of H minus
k << 45 (so that adding ki << 45, ki = 128 n + k, yields the bits of 2^n H).  picasso_amd/csrc/libm_glibc.h reads it.
"""
import struct
import sys
from decimal import Decimal, getcontext

getcontext().prec = 80


def bits(d):
    return struct.unpack("<Q", struct.pack("<d", d))[0]


def table():
    out = []
    for k in range(128):
        e = Decimal(2) ** (Decimal(k) / Decimal(128))
        h = float(e)                                   # float(Decimal) rounds to nearest
        t = float((e - Decimal(h)) / Decimal(h))
        out += [bits(t), (bits(h) - (k << 45)) & (2 ** 64 - 1)]
    return out


def main():
    tab = table()
    if len(sys.argv) > 2 and sys.argv[1] == "--check":
        blob = open(sys.argv[2], "rb").read()
        print("found" if struct.pack("<256Q", *tab) in blob else "NOT found")
        sys.exit(0)
    print("// 2^(k/128), k = 0 .. 127: bits of the tail T and of H - (k << 45); written by tools/gen_glibc_exp_table.py")
    for k in range(0, 256, 4):
        print("    " + " ".join(f"0x{v:016x}ull," for v in tab[k:k + 4]))


if __name__ == "__main__":
    main()
