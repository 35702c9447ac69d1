"""Static audit of the gfx950 code objects under latent-diffusion_amd/build: per kernel the register and LDS budget from the
code-object metadata and, from the disassembly, the things that silently cost a hot loop -- scratch traffic (spills),
waterfall loops (a buffer descriptor the compiler could not prove wave-uniform: v_readfirstlane + s_cbranch_execnz around the
load; a clamped `bytes - offset` did exactly that to the attention kernels in round 2), AGPR shuffling, matrix-core
instruction counts.  No GPU needed.     python tools/isa_audit.py [--markdown]
tests/test_isa_audit_cpu.py pins the invariants (no spills / no waterfall in the forward GEMM and attention kernels)."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BUILD = os.path.join(ROOT, "latent-diffusion_amd", "build")
LLVM = "/opt/rocm/lib/llvm/bin"


def code_object(obj, tmp):
    fat = os.path.join(tmp, "fat.bin")
    hsaco = os.path.join(tmp, "k.hsaco")
    # an explicit output file: without one llvm-objcopy rewrites its INPUT in place (and make then relinks the library)
    if subprocess.run([f"{LLVM}/llvm-objcopy", "--dump-section", f".hip_fatbin={fat}", obj, os.path.join(tmp, "copy.o")],
                      capture_output=True).returncode:
        return None
    r = subprocess.run([f"{LLVM}/clang-offload-bundler", "--type=o", "--unbundle", f"--input={fat}",
                        "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={hsaco}"], capture_output=True)
    return hsaco if r.returncode == 0 and os.path.getsize(hsaco) > 0 else None


def demangle(names):
    out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.splitlines()
    return dict(zip(names, out))


def audit(obj, tmp="/tmp"):
    hsaco = code_object(obj, tmp)
    if hsaco is None:
        return []
    notes = subprocess.run([f"{LLVM}/llvm-readelf", "--notes", hsaco], capture_output=True, text=True).stdout
    meta = {}
    for blk in notes.split("- .agpr_count:")[1:]:
        blk = ".agpr_count:" + blk
        f = {k: v for k, v in re.findall(r"\.(\w+):\s+(\S+)", blk)}
        if "name" in f:
            meta[f["name"]] = f
    dis = subprocess.run([f"{LLVM}/llvm-objdump", "-d", hsaco], capture_output=True, text=True).stdout
    rows = []
    for m in re.finditer(r"^[0-9a-f]+ <([^>]+)>:\n(.*?)(?=^[0-9a-f]+ <|\Z)", dis, re.S | re.M):
        name, body = m.group(1), m.group(2)
        if name not in meta:
            continue
        ops = re.findall(r"^\s+([a-z_0-9]+)", body, re.M)
        cnt = lambda p: sum(1 for o in ops if re.fullmatch(p, o))      # noqa: E731
        f = meta[name]
        # waterfall loop: v_readfirstlane ... s_and_saveexec ... buffer_* ... s_cbranch_execnz within a dozen instructions
        wf = 0
        for i, o in enumerate(ops):
            if o == "s_cbranch_execnz":
                w = ops[max(0, i - 14):i]
                if any(x.startswith("v_readfirstlane") for x in w) and any(x.startswith("buffer_") for x in w):
                    wf += 1
        rows.append(dict(symbol=name, vgpr=int(f.get("vgpr_count", 0)), agpr=int(f.get("agpr_count", 0)), sgpr=int(f.get("sgpr_count", 0)),
                         lds=int(f.get("group_segment_fixed_size", 0)), scratch=int(f.get("private_segment_fixed_size", 0)),
                         insts=len(ops), mfma=cnt(r"v_mfma_.*"), scratch_ops=cnt(r"scratch_(load|store).*"),
                         readfirstlane=cnt(r"v_readfirstlane_b32"), waterfall=wf,
                         accvgpr=cnt(r"v_accvgpr_(read|write).*"), dma=sum(1 for ln in body.splitlines() if "buffer_load" in ln and " lds" in ln)))
    names = demangle([r["symbol"] for r in rows])
    for r in rows:
        r["kernel"] = re.sub(r"\(.*", "", names.get(r["symbol"], r["symbol"])).replace("void ldm::", "").replace("ldm::", "")
    return rows


def all_rows():
    rows = []
    for f in sorted(os.listdir(BUILD)):
        if f.endswith(".hip.o"):
            for r in audit(os.path.join(BUILD, f)):
                r["file"] = f[:-6]
                rows.append(r)
    return rows


def main():
    rows = all_rows()
    cols = ["file", "kernel", "vgpr", "agpr", "sgpr", "lds", "scratch", "insts", "mfma", "dma", "scratch_ops", "readfirstlane", "waterfall", "accvgpr"]
    if "--markdown" in sys.argv:
        print("| " + " | ".join(cols) + " |")
        print("|" + "---|" * len(cols))
        for r in rows:
            print("| " + " | ".join(f"`{r[c]}`" if c == "kernel" else str(r[c]) for c in cols) + " |")
    else:
        for r in rows:
            print("  ".join(f"{c}={r[c]}" for c in cols))


if __name__ == "__main__":
    main()
