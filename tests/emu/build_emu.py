"""Compiles the kernel sources of latent-diffusion_amd/csrc for the HOST against the SIMT emulator (tests/emu/emu.h) ->
tests/emu/_build/libldm_emu.so, which exports the same C-ABI as libldm_amd.so and runs on CPU memory.  TEST
INFRASTRUCTURE: the product never loads it.

The sources are compiled as they are, after a textual pass over the constructs that have no host spelling:
  * ``extern __shared__ T name[];``            -> a pointer to the emulator's guarded LDS image
  * ``__attribute__((address_space(3)))``      -> dropped
  * ``asm volatile("s_waitcnt vmcnt(n)" ...)`` -> emu::wait_vmcnt(n) (retires queued LDS-DMA)
  * ``asm volatile("s_waitcnt lgkmcnt ...")``  -> a wave rendezvous (lanes exchange data through LDS around it)
  * empty asm fences / register pins           -> dropped
  * ``asm volatile("ds_read_b64_tr_b16 ...")`` -> the emulated transposed read
The transformed copies live only under _build/ (git-ignored)."""
import hashlib
import os
import re
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
CSRC = os.path.join(ROOT, "latent-diffusion_amd", "csrc")
OUT_DIR = os.path.join(HERE, "_build")
OUT = os.path.join(OUT_DIR, "libldm_emu.so")
CLANG = os.environ.get("LDM_EMU_CXX", "/opt/rocm/lib/llvm/bin/clang++")

_SHARED_RE = re.compile(r"extern\s+__shared__\s+(?:__attribute__\(\(aligned\(\d+\)\)\)\s+)?([\w ]+?)\s+(\w+)\[\];")


def _split_top(s, sep):
    """Split s on sep outside parentheses and string literals."""
    out, depth, cur, i, in_str = [], 0, [], 0, False
    while i < len(s):
        c = s[i]
        if in_str:
            cur.append(c)
            if c == "\\":
                cur.append(s[i + 1])
                i += 1
            elif c == '"':
                in_str = False
        elif c == '"':
            in_str = True
            cur.append(c)
        elif c in "([{":
            depth += 1
            cur.append(c)
        elif c in ")]}":
            depth -= 1
            cur.append(c)
        elif c == sep and depth == 0:
            out.append("".join(cur))
            cur = []
        else:
            cur.append(c)
        i += 1
    out.append("".join(cur))
    return out


def _operands(section):
    """'"=v"(x), "v"(y + 1)' -> ['x', 'y + 1']"""
    ops = []
    for part in _split_top(section, ","):
        part = part.strip()
        if not part:
            continue
        m = re.match(r'"[^"]*"\s*\((.*)\)\s*$', part, re.S)
        if not m:
            raise ValueError(f"asm operand not understood: {part!r}")
        ops.append(m.group(1).strip())
    return ops


def _asm_replacement(body, where):
    secs = _split_top(body, ":")
    tmpl = "".join(re.findall(r'"((?:[^"\\]|\\.)*)"', secs[0]))
    outs = _operands(secs[1]) if len(secs) > 1 else []
    ins = _operands(secs[2]) if len(secs) > 2 else []
    t = tmpl.strip()
    if t == "":
        return "((void)0)"
    m = re.fullmatch(r"s_waitcnt vmcnt\((%0|\d+)\)", t)
    if m:
        return f"::emu::wait_vmcnt({ins[0] if m.group(1) == '%0' else m.group(1)})"
    if re.fullmatch(r"s_waitcnt lgkmcnt\((%0|\d+)\)", t):
        # the kernels put this wait between LDS writes of some lanes and LDS reads of others in the same wave: on the
        # hardware the wave runs in lockstep, here the lanes are fibers and have to meet
        return "::emu::wave_barrier()"
    m = re.fullmatch(r"ds_read_b64_tr_b16 %0, %1(?: offset:(%2|\d+))?", t)
    if m:
        off = "0" if m.group(1) is None else (ins[1] if m.group(1) == "%2" else m.group(1))
        return f"::emu::ds_read_tr16_b64_lds_addr({outs[0]}, {ins[0]}, {off})"
    raise ValueError(f"{where}: inline asm not understood: {t!r}")


def translate(text, name):
    text = _SHARED_RE.sub(lambda m: f"{m.group(1)}* const {m.group(2)} = reinterpret_cast<{m.group(1)}*>(::emu::dyn_smem());", text)
    text = text.replace("__attribute__((address_space(3)))", "")
    out, i = [], 0
    key = "asm volatile("
    while True:
        j = text.find(key, i)
        if j < 0:
            out.append(text[i:])
            break
        out.append(text[i:j])
        k, depth, in_str = j + len(key), 1, False
        while depth:
            c = text[k]
            if in_str:
                if c == "\\":
                    k += 1
                elif c == '"':
                    in_str = False
            elif c == '"':
                in_str = True
            elif c == "(":
                depth += 1
            elif c == ")":
                depth -= 1
            k += 1
        out.append(_asm_replacement(text[j + len(key):k - 1], name))
        i = k
    return "".join(out)


def _sources():
    files = sorted(f for f in os.listdir(CSRC) if f.endswith(".hip")) + ["capi.cpp"]
    return [os.path.join(CSRC, f) for f in files] + [os.path.join(HERE, "selftest_kernels.cpp")]


def _stamp():
    h = hashlib.sha256()
    for p in _sources() + [os.path.join(CSRC, "common.h"), os.path.join(ROOT, "include", "ldm_amd.h"), os.path.join(HERE, "emu.h"),
                           os.path.join(HERE, "emu.cpp"), os.path.join(HERE, "shim", "hip", "hip_runtime.h"), os.path.abspath(__file__)]:
        with open(p, "rb") as f:
            h.update(f.read())
    try:      # -march=native: a library built on another host CPU must not be reused
        with open("/proc/cpuinfo") as f:
            h.update("".join(sorted({ln for ln in f if ln.startswith(("model name", "flags"))})).encode())
    except OSError:
        pass
    return h.hexdigest()


def build(force=False, verbose=False):
    """Build (if stale) and return the path of libldm_emu.so."""
    os.makedirs(OUT_DIR, exist_ok=True)
    stamp_file = os.path.join(OUT_DIR, "stamp")
    stamp = _stamp()
    if not force and os.path.exists(OUT) and os.path.exists(stamp_file) and open(stamp_file).read() == stamp:
        return OUT
    flags = ["-std=c++20", "-O2", "-g1", "-fPIC", "-w", "-fno-strict-aliasing", "-march=native", "-ffp-contract=fast",
             "-I", os.path.join(HERE, "shim"), "-I", HERE, "-I", CSRC]
    objs, procs = [], []
    for src in _sources():
        base = os.path.basename(src)
        cpp = os.path.join(OUT_DIR, base + ".emu.cpp")
        with open(src) as f:
            text = translate(f.read(), base)
        with open(cpp, "w") as f:
            f.write(f'#line 1 "{src}"\n' + text)
        obj = cpp + ".o"
        extra = ["-ffp-contract=off"] if base == "scheduler.hip" else []
        procs.append((base, subprocess.Popen([CLANG, *flags, *extra, "-x", "c++", "-c", cpp, "-o", obj],
                                             stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
        objs.append(obj)
    eobj = os.path.join(OUT_DIR, "emu.o")
    procs.append(("emu.cpp", subprocess.Popen([CLANG, "-std=c++20", "-O2", "-g1", "-fPIC", "-march=native", "-I", HERE, "-c", os.path.join(HERE, "emu.cpp"),
                                               "-o", eobj], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    objs.append(eobj)
    failed = False
    for base, p in procs:
        log = p.communicate()[0]
        if p.returncode != 0:
            failed = True
            sys.stderr.write(f"--- {base}\n{log[-6000:]}\n")
        elif verbose and log.strip():
            sys.stderr.write(f"--- {base}\n{log[-2000:]}\n")
    if failed:
        raise RuntimeError("emulator build failed")
    subprocess.run([CLANG, "-shared", "-fPIC", "-o", OUT, *objs], check=True)
    with open(stamp_file, "w") as f:
        f.write(stamp)
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
