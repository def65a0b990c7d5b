#!/usr/bin/env python
"""Build a tagged librlxhip whose DEVICE code of one source comes from hand-patched assembly (docs/PACKED_F32_HAZARD.md, round 5).

Inline asm in the C++ source changes what the compiler's unroller / scheduler / register allocator do, so it cannot test "the same
schedule plus one wait state".  This tool keeps the compiler's output: it compiles <source> to gfx950 assembly with the VECTORIZED
flags (the failing build), applies a patch to the text of ONE kernel, assembles and links the code object, bundles it and compiles
the host side of the same source around it -- the steps `hipcc -###` shows, with the assembly edited in between.

    python tools/probes/asm_patch_build.py <tag> <patch>        # -> rl-x_amd/lib/librlxhip_<tag>.so   (all other sources: default flags)

patches (k_head_bwd's two-column dZ loop, the block the round-4 bisection ended at):
    none        the compiler's assembly, unchanged (control: must fail like RLX_REPRO_PACKED_F32_FILES=sac.hip)
    nopN_both   `s_nop N` between the last packed FMA that reads the running sums as src2 and the ds_read_b128 that overwrite
                those registers -- at both places of the loop (middle, and back edge)
    nopN_mid / nopN_edge   only one of the two places
    wait_first  s_waitcnt lgkmcnt(0) in front of the FIRST packed FMA of each half (no FMA overlaps an LDS return)
    vnop_both   four v_nop instead of s_nop (the VALU itself drains)
    nosel       the second column's packed FMAs read w from a fresh register pair {w, w} (two v_mov) instead of selecting the high
                word of v[56:57] with op_sel:[0,1,0]
    fresh       the second column's ds_read_b128 land in fresh registers (v72..v87) instead of the registers the first column's
                packed FMAs have just read as src0 / src2 (no write-after-read left in the loop); kernel descriptor: 88 VGPRs
    nosel_fresh both
    selcopy     control for nosel: the second column keeps op_sel:[0,1,0] but reads it from a COPY of the pair (v[88:89] = v[56:57],
                two v_mov) -- separates "the modifier" from "the pair ds_read2_b32 has written"
"""
import os, re, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "rl-x_amd"))
LLVM = "/opt/rocm/lib/llvm/bin"
KERNEL = "_ZN3rlx10k_head_bwdEPfPKfS2_S0_liiiNS_4TwinE"


def run(cmd):
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode:
        raise SystemExit("FAILED: " + " ".join(cmd) + "\n" + r.stdout + r.stderr)
    return r


def patch_text(asm, patch):
    lines = asm.split("\n")
    a = next(i for i, l in enumerate(lines) if l.startswith(KERNEL + ":"))
    b = next(i for i in range(a, len(lines)) if lines[i].startswith("\t.size\t" + KERNEL) or ".Lfunc_end" in lines[i] and i > a + 10)
    # the loop: a basic block with 16 v_pk_fma_f32 and 8 ds_read_b128 ending in s_cbranch_scc0 to its own label
    pk = [i for i in range(a, b) if "v_pk_fma_f32" in lines[i]]
    loop = None
    for i in range(a, b):
        m = re.match(r"^(\.LBB\d+_\d+):", lines[i])
        if not m:
            continue
        j = next((k for k in range(i + 1, b) if lines[k].strip().startswith("s_cbranch") or re.match(r"^\.LBB", lines[k])), None)
        if j is None or m.group(1) not in lines[j]:
            continue
        body = range(i, j + 1)
        if sum("v_pk_fma_f32" in lines[k] for k in body) == 16 and sum("ds_read_b128" in lines[k] for k in body) == 8:
            loop = (i, j)
            break
    if loop is None:
        raise SystemExit(f"the two-column packed loop was not found in {KERNEL} ({len(pk)} v_pk_fma_f32 in the kernel)")
    i0, j0 = loop
    reads = [k for k in range(i0, j0) if "ds_read_b128" in lines[k]]
    mid = reads[4]                                   # first ds_read_b128 of the second column: lands in the addend registers
    edge = next(k for k in range(j0, i0, -1) if lines[k].strip().startswith("s_cmp"))     # after the last packed FMA of the body
    if patch == "none":
        ins = {}
    elif m := re.match(r"nop(\d+)_(both|mid|edge)$", patch):
        s = f"\ts_nop {int(m.group(1))}"
        ins = {k: [s] for k, w in ((mid, "mid"), (edge, "edge")) if m.group(2) in ("both", w)}
    elif patch == "vnop_both":
        ins = {mid: ["\tv_nop"] * 4, edge: ["\tv_nop"] * 4}
    elif patch == "wait_first":
        firsts = [next(k for k in range(i0, j0) if "v_pk_fma_f32" in lines[k]),
                  next(k for k in range(mid, j0) if "v_pk_fma_f32" in lines[k] and "op_sel:[0,1,0]" in lines[k])]
        ins = {k: ["\ts_waitcnt lgkmcnt(0)"] for k in firsts}
    elif patch in ("nosel", "fresh", "nosel_fresh", "selcopy"):
        ins = {}
        second = [k for k in range(mid, j0) if "v_pk_fma_f32" in lines[k] and "op_sel:[0,1,0]" in lines[k]]
        assert len(second) == 8, second
        wsrc = re.search(r"ds_read2_b32 v\[(\d+):(\d+)\]", "\n".join(lines[i0:j0])).groups()
        if patch == "selcopy":
            ins[second[0]] = [f"\tv_mov_b32_e32 v88, v{wsrc[0]}", f"\tv_mov_b32_e32 v89, v{wsrc[1]}"]
            for k in second:
                lines[k] = lines[k].replace(f"v[{wsrc[0]}:{wsrc[1]}]", "v[88:89]")
        if "nosel" in patch:
            ins[second[0]] = [f"\tv_mov_b32_e32 v88, v{wsrc[1]}", f"\tv_mov_b32_e32 v89, v{wsrc[1]}"]
            for k in second:
                lines[k] = lines[k].replace(f"v[{wsrc[0]}:{wsrc[1]}]", "v[88:89]").replace(" op_sel:[0,1,0]", "")
        if "fresh" in patch:
            remap = {}
            for n, k in enumerate(reads[4:]):
                lo, hi = map(int, re.search(r"ds_read_b128 v\[(\d+):(\d+)\]", lines[k]).groups())
                for q in range(4):
                    remap[lo + q] = 72 + 4 * n + q
                lines[k] = lines[k].replace(f"v[{lo}:{hi}]", f"v[{72 + 4 * n}:{75 + 4 * n}]")
            for k in second:       # src0 = the operand right after the destination
                m2 = re.match(r"(\s*v_pk_fma_f32 v\[\d+:\d+\], )v\[(\d+):(\d+)\](.*)", lines[k])
                a0 = int(m2.group(2))
                lines[k] = f"{m2.group(1)}v[{remap[a0]}:{remap[a0] + 1}]{m2.group(4)}"
        # the kernel's descriptor: room for v72..v89
        d0 = next(i for i, l in enumerate(lines) if l.strip() == ".amdhsa_kernel " + KERNEL)
        for i in range(d0, d0 + 80):
            if ".amdhsa_next_free_vgpr" in lines[i]:
                lines[i] = "\t\t.amdhsa_next_free_vgpr 90"
            if ".amdhsa_accum_offset" in lines[i]:
                lines[i] = "\t\t.amdhsa_accum_offset 92"
            if ".end_amdhsa_kernel" in lines[i]:
                break
    else:
        raise SystemExit("unknown patch " + patch)
    out = []
    for k, l in enumerate(lines):
        if k in ins:
            out += ins[k]
        out.append(l)
    body = "\n".join((ins.get(k, []) and "\n".join(ins[k]) + "\n" or "") + lines[k] for k in range(i0, j0 + 1))
    return "\n".join(out), body


def main():
    tag, patch = sys.argv[1], sys.argv[2]
    src = sys.argv[3] if len(sys.argv) > 3 else "sac.hip"
    os.environ["RLX_BUILD_TAG"] = tag
    import build as B                                                   # objects of every source with the DEFAULT flags
    B.build()
    spath, obj = os.path.join(B.CSRC, src), os.path.join(B.OBJ_DIR, src[:-4] + ".o")
    vec = [f for f in B.CFLAGS if f not in B._NOVEC]
    tmp = os.path.join(B.OBJ_DIR, "asm_" + src[:-4])
    run([B.HIPCC] + vec + ["--cuda-device-only", "-S", spath, "-o", tmp + ".s"])
    asm, body = patch_text(open(tmp + ".s").read(), patch)
    open(tmp + "_patched.s", "w").write(asm)
    open(os.path.join(B.LIB_DIR, f"loop_{tag}.s"), "w").write(body + "\n")
    run([LLVM + "/clang", "-x", "assembler", "-target", "amdgcn-amd-amdhsa", "-mcpu=gfx950", "-c", tmp + "_patched.s", "-o", tmp + "_dev.o"])
    run([LLVM + "/lld", "-flavor", "gnu", "-m", "elf64_amdgpu", "--no-undefined", "-shared", "-o", tmp + ".hsaco", tmp + "_dev.o"])
    run([LLVM + "/clang-offload-bundler", "-type=o", "-bundle-align=4096",
         "-targets=host-x86_64-unknown-linux-gnu,hipv4-amdgcn-amd-amdhsa--gfx950", "-input=/dev/null", "-input=" + tmp + ".hsaco",
         "-output=" + tmp + ".hipfb"])
    run([B.HIPCC] + vec + ["--cuda-host-only", "-Xclang", "-fcuda-include-gpubinary", "-Xclang", tmp + ".hipfb", "-c", spath, "-o", obj])
    objs = [os.path.join(B.OBJ_DIR, s[:-4] + ".o") for s in B._sources()]
    run([B.HIPCC, "--offload-arch=" + B.ARCH, "-shared", "-fPIC", "-o", B.LIB_PATH] + objs)
    print(B.LIB_PATH)


if __name__ == "__main__":
    main()
