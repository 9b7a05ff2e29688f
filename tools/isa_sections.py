"""Diagnostic: instruction classes per marked section of a kernel in an ISA listing made with -DCPT_ISA_MARKS
(hipcc -S --cuda-device-only -DCPT_ISA_MARKS cpt_perturb.hip -o x.s;  python tools/isa_sections.py x.s k_perturbILi1ELi0ELi0ELi1EE)."""
import sys

lines = open(sys.argv[1]).read().split("\n")
start = next(i for i, l in enumerate(lines) if sys.argv[2] in l and "@function" in l)
end = next(i for i in range(start, len(lines)) if "End function" in lines[i])
marks = [(i, lines[i].split("====")[1].strip()) for i in range(start, end) if "====" in lines[i]]
marks.append((end, "END"))
prev = (start, "FUNC")
for m in marks:
    ins = [l for l in lines[prev[0]:m[0]] if l.startswith("\t") and not l.strip().startswith((";", "."))]
    n = lambda key: sum(key in l for l in ins)
    print("%-14s -> %-14s instrs %5d  acc_read %4d acc_write %4d scratch %3d readlane %4d writelane %3d ds_read %3d nop %3d waitcnt %3d" % (
        prev[1], m[1], len(ins), n("v_accvgpr_read"), n("v_accvgpr_write"), n("scratch_"), n("v_readlane"), n("v_writelane"), n("ds_read"), n("s_nop"), n("s_waitcnt")))
    prev = m
for l in lines[start:end + 200]:
    if any(k in l for k in (".vgpr_count", ".agpr_count", "NumVgprs", "NumAgprs", "ScratchSize", "; Occupancy", ".private_segment_fixed_size", "; LDSByteSize")):
        print(l.strip())
