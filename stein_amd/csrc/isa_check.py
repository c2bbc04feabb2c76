"""Build-time audit of the gfx950 assembly of the kernels that stream memory through inline asm with hand-counted waits
(k_phi_x3fs in stein_x3.hip, k_distance_panel and k_distance_panel_deep in stein_dpanel.hip).  hipcc neither counts nor pads what is inside an
`asm` statement (cdna_hip_programming.md 5.7), so three things that it normally guarantees are checked here instead, on the
`.s` that -save-temps leaves beside the object; __graft_entry__.build() fails when one of them is violated:

  1. no scratch: `.vgpr_spill_count` = 0 and `.private_segment_fixed_size` = 0 (a spill or reload is a vector-memory
     operation nobody counted, and a reload can fetch a register whose streamed load has not landed yet);
  2. SGPR hazard: a vector-memory instruction inside an asm statement must not read, as its scalar base, an SGPR that a
     VALU instruction (v_readfirstlane / v_readlane -- e.g. the reload of a spilled SGPR -- or a compare) wrote fewer than
     5 wait states earlier; the statement's own leading `s_nop N` counts;
  3. a register that an asm load (or returning atomic) writes is not mentioned by any compiler-generated instruction
     while the load may be in flight: no copy, no move, no reuse.  Two strengths: for every kernel, the straight-line code
     behind the statement up to the next label, branch or asm `s_waitcnt vmcnt` (where a register-allocator copy of an asm
     output would sit); for the kernels named in PATH_PREFIXES, every control-flow path from the statement to an asm
     `s_waitcnt vmcnt` (their sources are written so that each such path passes one).

usage: python isa_check.py file.s [kernel-name-prefix ...]
"""
import re
import sys

DEFAULT_PREFIXES = ("_Z10k_phi_x3fs", "_Z16k_distance_panel", "_Z21k_distance_panel_deep")
PATH_PREFIXES = ("_Z16k_distance_panel", "_Z21k_distance_panel_deep")
_VMEM = re.compile(r"\b(global_load\w*|global_atomic\w*|global_store\w*|buffer_load\w*|buffer_store\w*)\b")
_SREG = re.compile(r"\bs\[(\d+):(\d+)\]|\bs(\d+)\b")
_VREG = re.compile(r"\bv\[(\d+):(\d+)\]|\bv(\d+)\b")


def _regs(pattern, text):
    out = set()
    for m in pattern.finditer(text):
        if m.group(1) is not None:
            out.update(range(int(m.group(1)), int(m.group(2)) + 1))
        else:
            out.add(int(m.group(3)))
    return out


def _instr(line):
    """the instruction text of a line, or None for labels / comments / directives"""
    t = line.split(";")[0].strip()
    if not t or t.endswith(":") or t.startswith("."):
        return None
    return t


def _functions(lines):
    """{name: (first line index, last line index)} of the code of every global function"""
    out, cur, start = {}, None, 0
    for i, ln in enumerate(lines):
        m = re.match(r"^([A-Za-z_][\w$.]*):\s*(;.*)?$", ln)
        if m and not m.group(1).startswith(".L") and not ln.startswith("\t"):
            if cur is not None:
                out[cur] = (start, i - 1)
            cur, start = m.group(1), i + 1
        if cur is not None and ln.startswith(".Lfunc_end"):
            out[cur] = (start, i - 1)
            cur = None
    return out


def _metadata(text):
    """{kernel name: {key: int}} from the amdhsa.kernels YAML at the end of the file"""
    meta, name = {}, None
    cur = {}
    for ln in text.splitlines():
        m = re.match(r"\s+(?:- )?\.(\w+):\s+(.*)$", ln)
        if not m:
            continue
        k, v = m.group(1), m.group(2).strip()
        if k == "name" and v.startswith("_Z"):
            name = v
            meta[name] = cur
            cur = {}
        elif k in ("vgpr_spill_count", "sgpr_spill_count", "private_segment_fixed_size", "vgpr_count"):
            try:
                cur[k] = int(v)
            except ValueError:
                pass
    return meta


def _kernel_meta(text, name):
    """the metadata block that contains `.name: <name>` (keys are sorted alphabetically: some precede the name)"""
    i = text.find(".name:           %s" % name)
    if i < 0:
        i = text.find(".name: %s" % name)
    if i < 0:
        m = re.search(r"\.name:\s+%s\s*$" % re.escape(name), text, re.M)
        if not m:
            return {}
        i = m.start()
    lo = text.rfind("  - .", 0, i)
    hi = text.find("\n  - .", i)
    block = text[lo:hi if hi > 0 else len(text)]
    out = {}
    for k in ("vgpr_spill_count", "sgpr_spill_count", "private_segment_fixed_size"):
        m = re.search(r"\.%s:\s+(\d+)" % k, block)
        if m:
            out[k] = int(m.group(1))
    return out


def check_text(text, prefixes=DEFAULT_PREFIXES):
    """-> list of violation strings (empty: clean)"""
    lines = text.splitlines()
    problems = []
    found = 0
    for name, (lo, hi) in _functions(lines).items():
        if not name.startswith(tuple(prefixes)):
            continue
        found += 1
        meta = _kernel_meta(text, name)
        for k in ("vgpr_spill_count", "private_segment_fixed_size"):
            if meta.get(k, 0) != 0:
                problems.append("%s: %s = %d (must be 0)" % (name, k, meta[k]))
        body = lines[lo:hi + 1]
        # asm blocks: (start, end) indices into body, with their instruction lines
        blocks, i = [], 0
        while i < len(body):
            if ";;#ASMSTART" in body[i]:
                j = i + 1
                while j < len(body) and ";;#ASMEND" not in body[j]:
                    j += 1
                blocks.append((i, j, [t for t in (_instr(x) for x in body[i + 1:j]) if t]))
                i = j
            i += 1
        in_asm = set()
        for a, b, _ in blocks:
            in_asm.update(range(a, b + 1))
        for a, b, ins in blocks:
            vm = [t for t in ins if _VMEM.search(t)]
            if not vm:
                continue
            # ---- 2. SGPR hazard ----
            sregs = set()
            for t in vm:
                sregs |= _regs(_SREG, t)
            pad = 0
            for t in ins:
                m = re.match(r"s_nop\s+(\d+)", t)
                if m:
                    pad += int(m.group(1)) + 1
                else:
                    break
            need, k = 5 - pad, a - 1
            while need > 0 and k >= 0:
                if k in in_asm:
                    k -= 1
                    continue
                t = _instr(body[k])
                if t is not None:
                    m = re.match(r"s_nop\s+(\d+)", t)
                    if m:
                        need -= int(m.group(1)) + 1
                    else:
                        if re.match(r"v_(readlane|readfirstlane)_b32|v_cmp\w*_e64|v_\w+_co_\w+_e64", t):
                            dst = _regs(_SREG, t.split(",")[0])
                            if dst & sregs:
                                problems.append("%s: line %d `%s` writes an SGPR that the asm vector-memory instruction `%s` "
                                                "reads %d wait state(s) later (needs 5: open the statement with s_nop 4)"
                                                % (name, lo + k + 1, t, vm[0], 5 - pad - need))
                        need -= 1
                k -= 1
            # ---- 3. destinations stay untouched until the next asm vmcnt wait ----
            dests = set()
            for t in vm:
                if re.match(r"(global_load|buffer_load)", t) or (re.match(r"global_atomic", t) and re.search(r"\bsc0\b|\bglc\b", t)):
                    dests |= _regs(_VREG, t.split(",")[0])
            if not dests:
                continue
            # follow the control flow from the statement: fall through, take unconditional branches, explore both sides of
            # conditional ones; a path ends at an asm `s_waitcnt vmcnt` (the registers may be used behind it) or at s_endpgm
            labels = {}
            for idx, ln in enumerate(body):
                m = re.match(r"^(\.LBB[\w]+):", ln)
                if m:
                    labels[m.group(1)] = idx
            all_paths = name.startswith(PATH_PREFIXES)
            work, seen, hit = [b + 1], set(), None
            while work and hit is None:
                k = work.pop()
                while k < len(body) and k not in seen:
                    seen.add(k)
                    if not all_paths and re.match(r"^\.LBB", body[k]):
                        break
                    if k in in_asm:
                        blk = next(x for x in blocks if x[0] <= k <= x[1])
                        if any(t.startswith("s_waitcnt") and "vmcnt" in t for t in blk[2]):
                            break
                        k = blk[1] + 1
                        continue
                    t = _instr(body[k])
                    if t is not None:
                        if _regs(_VREG, t) & dests:
                            hit = (k, t)
                            break
                        if t.startswith("s_endpgm"):
                            break
                        m = re.match(r"s_(c?branch)\w*\s+(\.LBB\w+)", t)
                        if m and not all_paths:
                            break
                        if m:
                            tgt = labels.get(m.group(2))
                            if t.startswith("s_branch"):
                                if tgt is None:
                                    break
                                k = tgt
                                continue
                            if tgt is not None:
                                work.append(tgt)
                    k += 1
            if hit is not None:
                problems.append("%s: line %d `%s` touches v%s while the asm load of line %d may still be in flight"
                                % (name, lo + hit[0] + 1, hit[1], sorted(_regs(_VREG, hit[1]) & dests), lo + a + 1))
    if not found:
        problems.append("no kernel with a name starting with %s in the file" % (prefixes,))
    return problems


def check_file(path, prefixes=DEFAULT_PREFIXES):
    with open(path) as f:
        return check_text(f.read(), prefixes)


if __name__ == "__main__":
    probs = check_file(sys.argv[1], tuple(sys.argv[2:]) or DEFAULT_PREFIXES)
    for p in probs:
        print(p)
    print("%d problem(s)" % len(probs))
    sys.exit(1 if probs else 0)
