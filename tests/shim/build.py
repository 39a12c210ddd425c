"""Builds the oracle-backed CLI (tests only): the host front end of better_flow_amd/host linked
against the C-ABI test shim instead of libbf_accel.so."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
OUT = os.path.join(HERE, "_build")


SAN = ["-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-fno-omit-frame-pointer", "-g"]


def build(sanitize=False):
    """sanitize: the same program under AddressSanitizer + UndefinedBehaviorSanitizer (sanitizers row):
    the oracle, the shim and the whole host front end in one instrumented binary."""
    os.makedirs(OUT, exist_ok=True)
    exe = os.path.join(OUT, "bf_motion_compensator_oracle" + ("_san" if sanitize else ""))
    srcs = [os.path.join(ROOT, "better_flow_amd", "host", "bf_motion_compensator.cpp"),
            os.path.join(HERE, "bf_accel_oracle_shim.cpp"), os.path.join(ROOT, "oracle", "bf_oracle.c")]
    import glob
    deps = srcs + glob.glob(os.path.join(ROOT, "better_flow_amd", "host", "better_flow", "*.h")) + \
        [os.path.join(ROOT, "include", "bf_accel.h"), os.path.join(ROOT, "oracle", "bf_oracle.h")]
    newest = max(os.path.getmtime(s) for s in deps)
    if os.path.exists(exe) and os.path.getmtime(exe) > newest:
        return exe
    extra = SAN if sanitize else []
    obj = os.path.join(OUT, "bf_oracle_san.o" if sanitize else "bf_oracle.o")
    subprocess.check_call(["gcc", "-O1" if sanitize else "-O2", "-std=c11", "-ffp-contract=off"] + extra + ["-c", srcs[2], "-o", obj])
    subprocess.check_call(["g++", "-O1" if sanitize else "-O2", "-std=c++14", "-pthread", "-ffp-contract=off"] + extra + [
                           "-I" + os.path.join(ROOT, "better_flow_amd", "host"), "-I" + os.path.join(ROOT, "include"),
                           srcs[0], srcs[1], obj, "-lm", "-o", exe])
    return exe


if __name__ == "__main__":
    print(build())
