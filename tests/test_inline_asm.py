"""Inline-assembly hygiene of the HIP sources (CPU only; runs `make asm`, ~30 s).

hipcc treats an `asm volatile(...)` statement as one opaque instruction: it does not know which status registers the
instructions inside write unless the clobber list says so.  Round 2 shipped, for a few commits, a forward kernel in which a
one-instruction statement (`s_and_b32`, which writes SCC) lacked the "scc" clobber: the compiler kept a comparison live in SCC
across it and one of two unrolled loop copies selected the wrong arm - 83 GPU tests stayed green.  This
test inspects every statement after a toolchain or source change:

  * source level: every instruction mnemonic inside every `asm volatile` template that writes SCC (scalar ALU / compare ops) or
    VCC (VOPC compares and carry-out adds in their implicit-VCC forms) must be matched by an "scc" / "vcc" clobber;
  * object level: `make asm` must produce, for every ;;#ASMSTART block of the generated assembly, instructions that one of the
    inspected source templates accounts for - a statement that reached the code object without passing this file's eyes fails.
"""
import glob
import os
import re
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "myslam_amd", "csrc")

# scalar ALU instructions that write SCC (CDNA3/4 ISA: SOP1 / SOP2 / SOPC / SOPK arithmetic, logic, compares, bit tests)
SCC_WRITERS = re.compile(
    r"^s_(and|or|xor|nand|nor|xnor|andn2|orn2|not|wqm|add|sub|addc|subb|min|max|abs|absdiff|lshl|lshr|ashr|bfe|bfm|lshl[1-4]_add|"
    r"cmp|cmpk|bitcmp0|bitcmp1|cmp_eq|cmp_lg|cmp_gt|cmp_ge|cmp_lt|cmp_le|bcnt0|bcnt1|ff0|ff1|flbit|quadmask|"
    r"and_saveexec|or_saveexec|xor_saveexec|andn2_saveexec|orn2_saveexec|nand_saveexec|nor_saveexec|xnor_saveexec|addk|mulk)"
    r"(_[a-z0-9]+)*$")
# vector instructions that write VCC implicitly in their e32 forms
VCC_WRITERS = re.compile(r"^v_(cmp|cmpx|add_co|sub_co|subrev_co|addc_co|subb_co|subbrev_co|div_scale|mad_u64_u32|mad_i64_i32)(_[a-z0-9]+)*$")


def _statements():
    """[(file, line, [mnemonics], clobbers set)] of every asm statement in the HIP sources."""
    out = []
    for path in sorted(glob.glob(os.path.join(CSRC, "*.hip")) + glob.glob(os.path.join(CSRC, "*.h"))):
        src = open(path).read()
        for m in re.finditer(r"\basm\s+volatile\s*\(", src):
            i, depth = m.end(), 1
            while depth:
                depth += {"(": 1, ")": -1}.get(src[i], 0)
                i += 1
            body = src[m.end():i - 1]
            strings = re.findall(r'"((?:[^"\\]|\\.)*)"', body)
            # template = the leading string literals (before the first ':' outside a string)
            head = re.split(r':(?=(?:[^"]*"[^"]*")*[^"]*$)', body)
            template = "".join(re.findall(r'"((?:[^"\\]|\\.)*)"', head[0]))
            clobbers = set(re.findall(r'"((?:[^"\\]|\\.)*)"', head[3])) if len(head) > 3 else set()
            mnems = [ln.strip().split()[0] for ln in re.split(r"\\n|\\t|;", template) if ln.strip()]
            out.append((os.path.basename(path), src.count("\n", 0, m.start()) + 1, mnems, clobbers, strings))
    return out


def test_every_inline_asm_statement_declares_the_status_registers_it_writes():
    stmts = _statements()
    assert len(stmts) >= 3, "the scan found no asm statements: has the source layout changed?"
    seen_scc = False
    for fname, line, mnems, clobbers, _ in stmts:
        for mn in mnems:
            if SCC_WRITERS.match(mn):
                seen_scc = True
                assert "scc" in clobbers, f"{fname}:{line}: `{mn}` writes SCC but the statement does not clobber \"scc\""
            if VCC_WRITERS.match(mn):
                assert "vcc" in clobbers, f"{fname}:{line}: `{mn}` writes VCC but the statement does not clobber \"vcc\""
    assert seen_scc, "opaque_zero()'s s_and_b32 was not found: the scan no longer sees eslam_common.h"


def test_scan_catches_a_missing_clobber(tmp_path, monkeypatch):
    """The scanner itself: the round-2 bug restated in a scratch file must be flagged."""
    bad = tmp_path / "bad.h"
    bad.write_text('void f(int& z) { asm volatile("s_and_b32 %0, %0, 0" : "+s"(z)); }\n'
                   'void g(int& z) { asm volatile("v_cmp_lt_f32 %0, %1\\n\\ts_nop 0" : : "v"(z), "v"(z) : "memory"); }\n')
    monkeypatch.setattr(globals()["glob"], "glob", lambda pat: [str(bad)] if pat.endswith("*.h") else [])
    stmts = _statements()
    assert [s[2] for s in stmts] == [["s_and_b32"], ["v_cmp_lt_f32", "s_nop"]]
    assert SCC_WRITERS.match("s_and_b32") and "scc" not in stmts[0][3]
    assert VCC_WRITERS.match("v_cmp_lt_f32") and "vcc" not in stmts[1][3]
    assert not SCC_WRITERS.match("s_waitcnt") and not SCC_WRITERS.match("s_nop") and not SCC_WRITERS.match("s_mov_b32")


def test_generated_assembly_holds_only_inspected_statements():
    """`make asm`: every ;;#ASMSTART block of every kernel's assembly consists of instructions the source scan has seen."""
    subprocess.check_call(["make", "-C", CSRC, "asm"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    known = set()
    for _, _, mnems, _, _ in _statements():
        known.update(mnems)
    blocks = 0
    for path in sorted(glob.glob(os.path.join(CSRC, "build", "asm", "*.s"))):
        text = open(path).read()
        for blk in re.findall(r";;#ASMSTART\n(.*?);;#ASMEND", text, flags=re.S):
            blocks += 1
            for ln in blk.splitlines():
                ln = ln.split(";")[0].strip()
                if ln:
                    assert ln.split()[0] in known, f"{os.path.basename(path)}: `{ln}` inside an asm block that no inspected source statement explains"
    assert blocks >= 10, blocks          # opaque_zero alone is inlined into dozens of loops
