"""Runs tools/mfma_ceiling.hip (bare fp32 MFMA loops, every CU, >= 2 s of back-to-back launches per
variant; see the header of that file) and prints its table.  Builds the binary first when it is
missing or older than its source.

    python tools/bench_mfma_ceiling.py [seconds per variant, default 2.5]
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SRC, BIN = os.path.join(HERE, "mfma_ceiling.hip"), os.path.join(HERE, "bin", "mfma_ceiling")


def main():
    if not os.path.exists(BIN) or os.path.getmtime(BIN) < os.path.getmtime(SRC):
        os.makedirs(os.path.dirname(BIN), exist_ok=True)
        subprocess.run(["hipcc", "-O3", "--offload-arch=gfx950", "-o", BIN, SRC], check=True)
    sys.exit(subprocess.run([BIN] + sys.argv[1:]).returncode)


if __name__ == "__main__":
    main()
