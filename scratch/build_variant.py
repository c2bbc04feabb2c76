"""Diagnostic builds of the library: python scratch/build_variant.py <name> [-DFLAG ...] -> scratch/lib_<name>.so
Recompiles the two translation units that hold the MFMA kernels with the extra flags and links them with the shipped
objects of the other two (run __graft_entry__.build() first).  Ablation / stamp builds only; nothing ships from here."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge
def main():
    name, flags = sys.argv[1], sys.argv[2:]
    audit = "--no-audit" not in flags          # timing-only ablations may break the audited invariants on purpose
    flags = [f for f in flags if f != "--no-audit"]
    ge.build()
    tmp = os.path.join(ge.OBJDIR, "variant_" + name)
    os.makedirs(tmp, exist_ok=True)
    objs, procs = [], []
    for src in ge.SRCS:
        base = os.path.basename(src)
        if base in ("stein_x3.hip", "steinhip.hip", "stein_dpanel.hip"):
            obj = os.path.join(tmp, base + ".o")
            audited = base in ge.ISA_CHECKED
            procs.append((src, subprocess.Popen([ge._hipcc()] + ge.HIPCC_FLAGS + flags + (["-save-temps=obj"] if audited else []) +
                                                ["-c", src, "-o", obj], stderr=subprocess.DEVNULL if audited else None)))
        else:
            obj = os.path.join(ge.OBJDIR, base + ".o")
        objs.append(obj)
    for src, p in procs:
        if p.wait(): raise SystemExit("compile failed")
        if audit: ge._isa_check(src, tmp)     # the same assembly audit as the shipped build (a variant's register pressure differs)
    out = os.path.join(ROOT, "scratch", "lib_%s.so" % name)
    subprocess.run([ge._hipcc()] + ge.LINK_FLAGS + objs + ["-o", out], check=True)
    print(out)
if __name__ == "__main__":
    main()
