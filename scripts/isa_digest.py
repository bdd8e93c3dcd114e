"""Per-kernel digest of the built library's gfx950 code: instruction count, VALU / LDS / VMEM counts and a hash of the
instruction stream (addresses and branch targets stripped), to prove that a source clean-up did not change a kernel:
   python scripts/isa_digest.py [substring] > before.txt ; ... ; python scripts/isa_digest.py [substring] | diff before.txt -
(SCN_LIB selects a variant build; --dump DIR also writes one .s file per kernel)"""
import hashlib, os, re, subprocess, sys, tempfile

lib = os.environ.get("SCN_LIB") or os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scanner_amd", "libscanner_hip.so")
args = [a for a in sys.argv[1:]]
dump = None
if "--dump" in args:
    k = args.index("--dump")
    dump = args[k + 1]
    del args[k:k + 2]
    os.makedirs(dump, exist_ok=True)
want = args[0] if args else ""
LLVM = "/opt/rocm/lib/llvm/bin"
rows = []
with tempfile.TemporaryDirectory() as d:
    subprocess.check_call(["objcopy", "-O", "binary", "--only-section=.hip_fatbin", lib, f"{d}/fat.bin"])
    blob = open(f"{d}/fat.bin", "rb").read()
    starts = [m.start() for m in re.finditer(b"__CLANG_OFFLOAD_BUNDLE__", blob)] + [len(blob)]
    for k in range(len(starts) - 1):
        open(f"{d}/b{k}.bin", "wb").write(blob[starts[k]:starts[k + 1]])
        r = subprocess.run([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950",
                            f"--input={d}/b{k}.bin", f"--output={d}/k{k}.co"], stderr=subprocess.DEVNULL)
        if r.returncode or not os.path.getsize(f"{d}/k{k}.co"):
            continue
        txt = subprocess.check_output([f"{LLVM}/llvm-objdump", "-d", "--no-show-raw-insn", f"{d}/k{k}.co"], text=True)
        name, body = None, []

        def flush():
            if name is None or not body:
                return
            dem = subprocess.check_output(["c++filt", name], text=True).strip()
            if want not in dem:
                return
            ins = [re.sub(r"\s*//.*$", "", l).strip() for l in body]
            ins = [re.sub(r"\b[0-9a-f]{8,16} <[^>]+>", "<target>", i) for i in ins if i]
            h = hashlib.sha256("\n".join(ins).encode()).hexdigest()[:12]
            op = [i.split()[0] for i in ins]
            valu = sum(o.startswith("v_") for o in op)
            rows.append((dem, len(ins), valu, sum(o.startswith("ds_") for o in op), sum(o.startswith(("buffer_", "global_", "flat_")) for o in op),
                         sum(o == "s_barrier" for o in op), h))
            if dump:
                open(os.path.join(dump, re.sub(r"[^A-Za-z0-9_]+", "_", dem)[:180] + ".s"), "w").write("\n".join(ins) + "\n")

        for line in txt.splitlines():
            m = re.match(r"^[0-9a-f]+ <([^>]+)>:$", line)
            if m:
                flush()
                name, body = m.group(1), []
            elif name is not None and line[:1] in " \t":
                body.append(line)
        flush()
for dem, n, valu, lds, vmem, bar, h in sorted(rows):
    print(f"{h} ins {n:6d} valu {valu:6d} lds {lds:5d} vmem {vmem:4d} bar {bar:3d}  {dem[:170]}")
