#!/usr/bin/env python3
"""Writes fusion-cryptography_amd/csrc/fz_keccak_x64.inc: the SHA-3 / SHAKE block loop (absorb a block or not, Keccak-f[1600],
emit a block or not) for x86-64 with BMI1/BMI2 as ONE assembly routine.

Why assembly.  hash_ag (fusion/fusion.py:632-652) is one serial SHAKE-256 over ~13.5 KB of text per signer, so the
permutation's time on ONE host core is the floor of aggregate() and verify().  Measured on the GPU box's Zen 5 core
(tools/microbench/keccak_host.cpp, profiles/r06_keccak_variants_gpu_host.txt): the compilers' register-resident C form takes
35 cycles per round, and so does a spill-free memory-resident form with 40 fewer instructions -- because neither is bound by
instruction count.  A round is  [theta: needs EVERY output of the previous round]  ->  [25 lanes of rho / pi / chi: ~180
instructions]  and the two phases do not overlap: the column parities wait for the last chi output (through a store and a
load in the memory form), D waits for the parities, every lane waits for D.  This routine shortens the serial part to four
instructions:
  * EARLY PARITY: every chi output is XORed into its column's running parity as it is produced (row 0 is computed straight
    into the parity registers), so C is complete one instruction after the last output;
  * the state lives in two 200-byte frames on the stack (a round reads one and writes the other), which frees the two
    pointer registers; round constant pointer, counters and arguments live in the frame too;
  * registers: b0..b4 (one row), C'0..C'4 (next round's parities), D0..D3, one temporary = all fifteen; D4 goes through the
    frame (its five consumers sit in the middle of the round, not at its start).
  * the block loop is inside: the state is copied in and out once per CALL, not once per block.
Every row is checked against the C permutation at load time (fz_host.cpp) and against hashlib in tests/test_host_pipeline.py.
    python tools/gen_keccak_x64.py            # rewrites the .inc (tests/test_host_pipeline.py checks it is up to date)
"""
import os
import sys

RHO = [0, 1, 62, 28, 27, 36, 44, 6, 55, 20, 3, 10, 43, 25, 39, 41, 45, 15, 21, 8, 18, 2, 61, 56, 14]      # index x + 5 y
B = ["%rax", "%rbx", "%rcx", "%rdx", "%rbp"]
C = ["%r8", "%r9", "%r10", "%r11", "%r12"]
D = ["%r13", "%r14", "%r15", "%rdi", None]               # D4 lives in the frame
T = "%rsi"
X, Y = 0, 200                                            # the two state frames
O_D4, O_RC, O_CNT, O_S, O_IN, O_OUT, O_NB = 400, 408, 416, 424, 432, 440, 448
FRAME = 472                                              # 6 pushes + return address + 472 = a multiple of 16
RATE_WORDS = 17


def rows():
    """output row y': B[x'][y'] = rol(A[x][y] ^ D[x], rho[x][y]) with x = (x' + 3 y') % 5, y = x'      (pi inverted)"""
    out = []
    for yy in range(5):
        row = []
        for xx in range(5):
            x, y = (xx + 3 * yy) % 5, xx
            row.append((x + 5 * y, x, RHO[x + 5 * y]))
        out.append(row)
    return out


VROWS = tuple(int(c) for c in os.environ.get("GEN_KECCAK_VROWS", "34"))      # the output rows the hybrid form computes in xmm registers
BCAST_XOR = os.environ.get("GEN_KECCAK_BCAST", "1") == "1"
XD = [f"%xmm{i}" for i in range(5)]                      # D[0..4] copies
XB = [f"%xmm{5 + i}" for i in range(5)]                  # one row of B
XE = [f"%xmm{10 + i}" for i in range(5)]                 # the vector rows' column parities
XT = "%xmm15"


def d_from_c(hybrid):
    L = []
    for x in range(4):
        L.append(f"rorx $63, {C[(x + 1) % 5]}, {D[x]}")
        L.append(f"xor {C[(x + 4) % 5]}, {D[x]}")
        if hybrid:
            L.append(f"vmovq {D[x]}, {XD[x]}")
    L.append(f"rorx $63, {C[0]}, {T}")
    L.append(f"xor {C[3]}, {T}")
    L.append(f"mov {T}, {O_D4}(%rsp)")
    if hybrid:
        L.append(f"vmovq {T}, {XD[4]}")
    return L


def vector_row(src, dst, yy, row, first):
    """one output row in xmm registers: vpxorq with the lane as an 8-byte broadcast memory operand (a 16-byte load would
    straddle two 8-byte stores of the previous round and could not be forwarded: 190 ns instead of 134), vprolq, chi as one
    vpternlogq, stores to the frame; the vector rows' column parities are collected in XE (XORing the stored outputs into
    the scalar parities from the frame instead measured 142 ns against 134)"""
    L = []
    for k in range(5):
        i, x, n = row[k]
        if BCAST_XOR:
            L.append(f"vpxorq {src + 8 * i}(%rsp){{1to2}}, {XD[x]}, {XB[k]}")
        else:
            L.append(f"vmovq {src + 8 * i}(%rsp), {XB[k]}")
            L.append(f"vpxor {XD[x]}, {XB[k]}, {XB[k]}")
        if n:
            L.append(f"vprolq ${n}, {XB[k]}, {XB[k]}")
    for k in range(5):
        # b[k] is also read by outputs k - 1 and k - 2: outputs 2, 3, 4 come after both and may overwrite it
        inplace = k >= 2
        tgt = XB[k] if inplace else (XE[k] if first else XT)
        if not inplace:
            L.append(f"vmovdqa {XB[k]}, {tgt}")
        L.append(f"vpternlogq $0xD2, {XB[(k + 2) % 5]}, {XB[(k + 1) % 5]}, {tgt}")      # b[k] ^ (~b[k+1] & b[k+2])
        L.append(f"vmovq {tgt}, {dst + 8 * (5 * yy + k)}(%rsp)")
        if first and inplace:
            L.append(f"vmovdqa {tgt}, {XE[k]}")
        if not first:
            L.append(f"vpxor {tgt}, {XE[k]}, {XE[k]}")
    return L


def scalar_row(src, dst, yy, row, into_parities):
    L = []
    order = sorted(range(5), key=lambda k: row[k][1] == 4)                      # the lane of column 4 (D4 comes from the frame) last
    for k in order:
        i, x, n = row[k]
        L.append(f"mov {src + 8 * i}(%rsp), {B[k]}")
        L.append(f"xor {D[x]}, {B[k]}" if D[x] else f"xor {O_D4}(%rsp), {B[k]}")
        if n:
            L.append(f"rorx ${64 - n}, {B[k]}, {B[k]}")
    for k in range(5):
        tgt = C[k] if into_parities else T
        L.append(f"andn {B[(k + 2) % 5]}, {B[(k + 1) % 5]}, {tgt}")              # ~b[k+1] & b[k+2]
        L.append(f"xor {B[k]}, {tgt}")
        if yy == 0 and k == 0:                                                  # iota (row 0 is always a scalar row)
            L.append(f"mov {O_RC}(%rsp), {T}")
            L.append(f"xor ({T}), {tgt}")
        L.append(f"mov {tgt}, {dst + 8 * (5 * yy + k)}(%rsp)")
        if not into_parities:
            L.append(f"xor {T}, {C[k]}")
    return L


def one_round(src, dst, hybrid=False):
    R = rows()
    vec = [r for r in range(5) if hybrid and r in VROWS]
    sca = [r for r in range(5) if r not in vec]
    L = []
    for n in range(max(len(vec), len(sca))):                                    # vector rows first, between the scalar ones
        if n < len(vec):
            L += vector_row(src, dst, vec[n], R[vec[n]], n == 0)
        if n < len(sca):
            L += scalar_row(src, dst, sca[n], R[sca[n]], n == 0)                # the first scalar row is computed INTO the parity registers
    if hybrid:                                                                  # the vector rows' parities join the scalar ones
        for k in range(5):
            L.append(f"vmovq {XE[k]}, {T}")
            L.append(f"xor {T}, {C[k]}")
    L.append(f"addq $8, {O_RC}(%rsp)")
    L += d_from_c(hybrid)
    return L


def routine(hybrid=False):
    name = "fz_keccak_blocks_x64v" if hybrid else "fz_keccak_blocks_x64"
    L = [".text", ".p2align 6", f".type {name},@function", f"{name}:",
         "push %rbx", "push %rbp", "push %r12", "push %r13", "push %r14", "push %r15",
         f"sub ${FRAME}, %rsp",
         f"mov %rdi, {O_S}(%rsp)", f"mov %rsi, {O_IN}(%rsp)", f"mov %rdx, {O_OUT}(%rsp)", f"mov %rcx, {O_NB}(%rsp)"]
    for i in range(25):
        L += [f"mov {8 * i}(%rdi), %rax", f"mov %rax, {X + 8 * i}(%rsp)"]
    L += ["test %rcx, %rcx", "jz 9f", ".p2align 5", "2:"]
    # absorb one block (if there is input)
    L += [f"mov {O_IN}(%rsp), {T}", f"test {T}, {T}", "jz 3f"]
    for i in range(RATE_WORDS):
        L += [f"mov {8 * i}({T}), %rax", f"xor %rax, {X + 8 * i}(%rsp)"]
    L += [f"addq ${8 * RATE_WORDS}, {O_IN}(%rsp)", "3:"]
    # column parities of the state, D, round constants, counter
    for x in range(5):
        L.append(f"mov {X + 8 * x}(%rsp), {C[x]}")
        for y in range(1, 5):
            L.append(f"xor {X + 8 * (x + 5 * y)}(%rsp), {C[x]}")
    L += d_from_c(hybrid)
    L += [f"lea fz_keccak_rc_x64(%rip), {T}", f"mov {T}, {O_RC}(%rsp)", f"movl $12, {O_CNT}(%rsp)", ".p2align 5", "1:"]
    L += one_round(X, Y, hybrid)
    L += one_round(Y, X, hybrid)
    L += [f"decl {O_CNT}(%rsp)", "jnz 1b"]
    # emit one block (if there is an output)
    L += [f"mov {O_OUT}(%rsp), {T}", f"test {T}, {T}", "jz 4f"]
    for i in range(RATE_WORDS):
        L += [f"mov {X + 8 * i}(%rsp), %rax", f"mov %rax, {8 * i}({T})"]
    L += [f"addq ${8 * RATE_WORDS}, {O_OUT}(%rsp)", "4:", f"decq {O_NB}(%rsp)", "jnz 2b", "9:", f"mov {O_S}(%rsp), %rdi"]
    for i in range(25):
        L += [f"mov {X + 8 * i}(%rsp), %rax", f"mov %rax, {8 * i}(%rdi)"]
    L += [f"add ${FRAME}, %rsp", "pop %r15", "pop %r14", "pop %r13", "pop %r12", "pop %rbp", "pop %rbx", "ret",
          f".size {name}, .-{name}"]
    return L


def text():
    body = routine(False) + routine(True)
    n_round = len(one_round(X, Y))
    n_hyb = len(one_round(X, Y, True))
    vrows_text = " and ".join(str(r) for r in VROWS)
    out = ["// GENERATED by tools/gen_keccak_x64.py -- do not edit.  SHA-3 / SHAKE block loop, x86-64 + BMI1/BMI2: early parity, the state",
           f"// in two stack frames, {n_round} instructions per round (see the generator for the register plan and the reasoning).",
           "// void fz_keccak_blocks_x64(uint64_t s[25], const uint8_t *in /* or NULL */, uint8_t *out /* or NULL */, size_t nblocks):",
           "//     nblocks times { s[0..16] ^= the next 136 bytes of in (if in); Keccak-f[1600](s); the next 136 bytes of out = s[0..16] (if out) }",
           "// reads the round constants from fz_keccak_rc_x64[24]",
           f"// fz_keccak_blocks_x64v: the same contract, AVX-512VL as well: output rows {vrows_text} of every round are computed in xmm registers",
           f"// ({n_hyb} instructions per round, a third of them on the vector pipes)",
           "__asm__("]
    for ln in body:
        out.append('    "' + ln + '\\n"')
    out.append(");")
    return "\n".join(out) + "\n"


if __name__ == "__main__":
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    path = os.path.join(root, "fusion-cryptography_amd", "csrc", "fz_keccak_x64.inc")
    if "--stdout" in sys.argv:
        sys.stdout.write(text())
        sys.exit(0)
    if "--check" in sys.argv:
        sys.exit(0 if open(path).read() == text() else 1)
    with open(path, "w") as fh:
        fh.write(text())
    print("wrote", path)
