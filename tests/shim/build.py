"""Builds the oracle-backed CLI (tests only): the host front end of better_flow_amd/host linked
against the C-ABI test shim instead of libbf_accel.so."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
OUT = os.path.join(HERE, "_build")


def build():
    os.makedirs(OUT, exist_ok=True)
    exe = os.path.join(OUT, "bf_motion_compensator_oracle")
    srcs = [os.path.join(ROOT, "better_flow_amd", "host", "bf_motion_compensator.cpp"),
            os.path.join(HERE, "bf_accel_oracle_shim.cpp"), os.path.join(ROOT, "oracle", "bf_oracle.c")]
    newest = max(os.path.getmtime(s) for s in srcs)
    if os.path.exists(exe) and os.path.getmtime(exe) > newest:
        return exe
    obj = os.path.join(OUT, "bf_oracle.o")
    subprocess.check_call(["gcc", "-O2", "-std=c11", "-ffp-contract=off", "-c", srcs[2], "-o", obj])
    subprocess.check_call(["g++", "-O2", "-std=c++14", "-ffp-contract=off",
                           "-I" + os.path.join(ROOT, "better_flow_amd", "host"), "-I" + os.path.join(ROOT, "include"),
                           srcs[0], srcs[1], obj, "-lm", "-o", exe])
    return exe


if __name__ == "__main__":
    print(build())
