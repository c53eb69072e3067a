"""tools/scan_store_waits.py -- dev-only, runs on the CPU (hipcc cross-compiles): compiles each .hip source to gfx950
assembly and lists the kernels that contain `s_waitcnt vmcnt(0)` BETWEEN their first and their last global store.

Why: vmcnt counts loads and stores together and retires in order.  A store under a lane mask sits behind an
s_cbranch_execz, so the compiler cannot count it, and a load whose first use comes after such a store is waited
for with vmcnt(0) -- the wave then waits for the ACKNOWLEDGEMENT of the stores it has just issued (DESIGN.md 4.1,
"Waits the compiler adds").  The cure is to pin what was loaded in registers before the first store:
    asm volatile("" : "+v"(x));
Grid-stride loops (load, store, next load) and explicit end-of-loop waits are reported too; read the listing.

    python tools/scan_store_waits.py [source.hip ...]        (default: every kernel source of the library)
"""
import os, re, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "mms_answer_selection_amd", "csrc")
sys.path.insert(0, ROOT)
from mms_answer_selection_amd import build  # noqa: E402


def scan(src):
    flags = [f for f in build.HIPCC_FLAGS if f not in ("-shared", "-fPIC")] + build.EXTRA_FLAGS.get(os.path.basename(src), [])
    with tempfile.NamedTemporaryFile(suffix=".s") as out:
        subprocess.check_call([build._hipcc()] + flags + ["-S", "--cuda-device-only", "-I", os.path.join(ROOT, "include"),
                                                          "-I", CSRC, src, "-o", out.name], stderr=subprocess.DEVNULL)
        name, lines = None, []
        for ln in open(out.name):
            m = re.match(r"^(_Z\w+):", ln)
            if m:
                name, lines = m.group(1), []
                continue
            if name is None:
                continue
            lines.append(ln)
            if "s_endpgm" in ln:
                st = [i for i, l in enumerate(lines) if "global_store" in l or "buffer_store" in l]
                w = [i for i, l in enumerate(lines) if re.search(r"s_waitcnt.*vmcnt\(0\)", l)]
                bad = [i for i in w if st and st[0] < i < st[-1]]
                if bad:
                    d = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
                    print("%-24s %s\n    vmcnt(0) at asm lines %s; stores span %d..%d" % (
                        os.path.basename(src), re.sub(r"\(.*", "", d)[:120], bad[:6], st[0], st[-1]))
                name = None


if __name__ == "__main__":
    srcs = sys.argv[1:] or [os.path.join(CSRC, f) for f in build.HIP_SOURCES]
    for s in srcs:
        scan(s)
