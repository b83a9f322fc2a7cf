"""In-tree build of libplenvec.so with hipcc for gfx950 (MI355X).  hipcc cross-compiles without a GPU."""
import os
import shutil
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
SRC = os.path.join(CSRC, "plenvec.hip")
OUT = os.path.join(CSRC, "libplenvec.so")
DEPS = [SRC, os.path.join(CSRC, "plen_model_gen.h"), os.path.join(CSRC, "plen_motor_pass_gen.h"), os.path.join(os.path.dirname(_HERE), "include", "plenvec.h")]


FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fno-slp-vectorize", "-fPIC", "-shared"]


def hipcc_path():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", shutil.which("hipcc")):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found (ROCm 7.x expected under /opt/rocm)")


RESOURCES = os.path.join(CSRC, "libplenvec.resources.json")


def parse_resource_remarks(text):
    """The compiler's -Rpass-analysis=kernel-resource-usage remarks as {kernel symbol: {"VGPRs": .., "SGPRs Spill": .., "ScratchSize": .., "Occupancy": .., "LDS Size": ..}}."""
    import re
    out, cur = {}, None
    for line in text.splitlines():
        m = re.search(r"remark:\s+Function Name: (\S+)", line)
        if m:
            cur = out.setdefault(m.group(1), {})
            continue
        m = re.search(r"remark:\s+([A-Za-z ]+?)(?: \[[^\]]*\])?: (\S+) \[-Rpass-analysis", line)
        if m and cur is not None:
            v = m.group(2)
            cur[m.group(1).strip()] = int(v) if v.lstrip("-").isdigit() else v
    return out


def build_extension(force=False, verbose=False):
    """Compile plenvec.hip -> csrc/libplenvec.so (gfx950 only).  Returns the output path.  The compiler's resource report of every kernel (registers, spilled SGPRs,
    scratch, LDS, occupancy) is kept next to the library as libplenvec.resources.json: tests/test_cabi_cpu.py holds the env kernels to their budgets."""
    import json
    if not force and os.path.exists(OUT) and os.path.exists(RESOURCES) and all(os.path.getmtime(OUT) >= os.path.getmtime(d) for d in DEPS):
        return OUT
    # -fno-slp-vectorize: the SLP vectoriser pairs the 3x3 kinematics products into v_pk_* ops but pays for it with more
    # v_mov shuffles than it saves (measured: -200 VALU instructions per substep, -3.6 % step time, 13 -> 5 spilled VGPRs);
    # the packed Delassus build uses explicit vector types and is unaffected.
    cmd = [hipcc_path(), "-Rpass-analysis=kernel-resource-usage"] + FLAGS + ["-o", OUT, SRC]
    r = subprocess.run(cmd, cwd=CSRC, stderr=subprocess.PIPE, text=True)
    if r.returncode != 0:
        raise RuntimeError("hipcc failed:\n" + r.stderr[-4000:])
    if verbose:
        print(r.stderr)
    with open(RESOURCES, "w") as f:
        json.dump(parse_resource_remarks(r.stderr), f, indent=1, sort_keys=True)
    return OUT


TD3_SRC = os.path.join(CSRC, "td3_kernels.hip")
TD3_OUT = os.path.join(CSRC, "libplentd3.so")


def build_td3_kernels(force=False):
    """Compile td3_kernels.hip -> csrc/libplentd3.so (gfx950): the fused TD3 update's non-GEMM kernels (include/plentd3.h)."""
    deps = [TD3_SRC, os.path.join(CSRC, "td3_rows.hip"), os.path.join(CSRC, "td3_team.hip"), os.path.join(CSRC, "td3_block.hip"), os.path.join(os.path.dirname(_HERE), "include", "plentd3.h")]
    if not force and os.path.exists(TD3_OUT) and all(os.path.getmtime(TD3_OUT) >= os.path.getmtime(d) for d in deps):
        return TD3_OUT
    subprocess.check_call([hipcc_path(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-o", TD3_OUT, TD3_SRC], cwd=CSRC)
    return TD3_OUT


# The A/B partner of the shipped env library (csrc/variants/nospec.so; tests/test_env_gpu.py holds the two to bitwise equality): every round-4/5 restructuring switched off --
# run-time point tests instead of the count-specialised solver loops (with their hoisted commits), the Delassus matrix as vector multiply-adds from broadcast LDS reads instead of
# matrix-core tiles.  (Not the mass matrix: -DPLENVEC_MFMA_MASS=0 changes how the compiler contracts the rest of that phase -- even the bias force, which the switch does not
# touch, moves in its last bit -- so that pair of builds is equal to rounding, not to the bit: scripts/gpu_twin_check.py.)
REFERENCE_FORM_FLAGS = ["-DPLENVEC_COUNT_SPECIALISED=0", "-DPLENVEC_MFMA_DELASSUS=0"]


def build_variant(name, defines=()):
    """Profiling / A-B builds (scripts/): csrc/variants/<name>.so with extra -D flags, rebuilt when the source is newer."""
    out = os.path.join(CSRC, "variants", name + ".so")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    if not (os.path.exists(out) and all(os.path.getmtime(out) >= os.path.getmtime(d) for d in DEPS)):
        subprocess.check_call([hipcc_path()] + FLAGS + list(defines) + ["-o", out, SRC], cwd=CSRC)
    return out
