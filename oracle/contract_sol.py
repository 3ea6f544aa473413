"""ORACLE — TEST INFRASTRUCTURE ONLY. A statement-by-statement Python restatement of
    /root/reference/solidity_verifier_contract/contract.sol  Halo2Verifier.verifyProof  (lines 62-828)
— the halo2-solidity-verifier output checked into the reference for its SquareCircuit
(src/signal.rs). This is the one artefact in the reference that pins the PROTOCOL layer of the hot
path: Keccak transcript order, evaluation layout (calldata offsets), the quotient identity, the
SHPLONK rotation sets {x}, {w^-6 x, x, w x}, {x, w x} with their zeta/nu powers, and the final check.

EVM model: `mem` is word-addressed memory (dict offset -> int), `calldataload` reads the ABI calldata
of verifyProof(address vk, bytes proof, uint256[] instances); the VK contract the original reads
with extcodecopy is absent from the reference, so the caller supplies the same words
(vk_words: list of 29 ints, layout contract.sol:14-35 followed by the fixed and permutation
commitments). Precompiles: 0x05 modexp -> pow, 0x06/0x07 ecAdd/ecMul -> affine arithmetic of
oracle/pyref.py. 0x08 (pairing) cannot be evaluated without an Fq12 tower; with the test SRS's known
trapdoor s the check e(lhs, G2) * e(rhs, -s G2) == 1 is equivalent to lhs == s * rhs in G1, which is
what `verify_proof` returns (the G2 words of the vk are therefore unused).

Every block below carries the line range of contract.sol it restates.
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import pyref as P  # noqa: E402
from plonk_ref import keccak256  # noqa: E402

# contract.sol:6-59
PROOF_LEN_CPTR, PROOF_CPTR, NUM_INSTANCE_CPTR, INSTANCE_CPTR = 0x64, 0x84, 0x04E4, 0x0504
FIRST_QUOTIENT_X_CPTR, LAST_QUOTIENT_X_CPTR = 0x0204, 0x0244
VK_MPTR = VK_DIGEST_MPTR = 0x0480
NUM_INSTANCES_MPTR, K_MPTR, N_INV_MPTR, OMEGA_MPTR, OMEGA_INV_MPTR, OMEGA_INV_TO_L_MPTR = 0x04A0, 0x04C0, 0x04E0, 0x0500, 0x0520, 0x0540
HAS_ACCUMULATOR_MPTR, G1_X_MPTR, G1_Y_MPTR = 0x0560, 0x05E0, 0x0600
CHALLENGE_MPTR = THETA_MPTR = 0x0820
BETA_MPTR, GAMMA_MPTR, Y_MPTR, X_MPTR, ZETA_MPTR, NU_MPTR, MU_MPTR = 0x0840, 0x0860, 0x0880, 0x08A0, 0x08C0, 0x08E0, 0x0900
X_N_MPTR, X_N_MINUS_1_INV_MPTR, L_LAST_MPTR, L_BLIND_MPTR, L_0_MPTR = 0x09A0, 0x09C0, 0x09E0, 0x0A00, 0x0A20
INSTANCE_EVAL_MPTR, QUOTIENT_EVAL_MPTR, QUOTIENT_X_MPTR, QUOTIENT_Y_MPTR, G1_SCALAR_MPTR = 0x0A40, 0x0A60, 0x0A80, 0x0AA0, 0x0AC0
PAIRING_LHS_X_MPTR, PAIRING_LHS_Y_MPTR, PAIRING_RHS_X_MPTR, PAIRING_RHS_Y_MPTR = 0x0AE0, 0x0B00, 0x0B20, 0x0B40

q = 21888242871839275222246405745257275088696311157297823662689037894645226208583  # :210
r = 21888242871839275222246405745257275088548364400416034343698204186575808495617  # :211


def make_calldata(proof, instances):
    """ABI encoding of verifyProof(address, bytes, uint256[]): selector, 3 head words, then the proof
    (length word at 0x64, bytes at 0x84) and the instances (length at 0x04e4, words at 0x0504) —
    the constants at contract.sol:6-9 fix this layout for a 0x460-byte proof."""
    cd = bytearray(4 + 3 * 32)
    cd += len(proof).to_bytes(32, "big") + bytes(proof)
    assert len(cd) == NUM_INSTANCE_CPTR, "proof length does not match the contract's calldata layout"
    cd += len(instances).to_bytes(32, "big")
    for v in instances:
        cd += int(v).to_bytes(32, "big")
    return bytes(cd)


def vk_words(vk_digest, num_instances, k, fixed_commitments, permutation_commitments):
    """The words extcodecopy(vk, VK_MPTR, 0, 0x3a0) would deliver (contract.sol:14-35,307)."""
    n = 1 << k
    omega = P.omega(k)
    omega_inv = pow(omega, -1, r)
    w = [vk_digest, num_instances, k, pow(n, -1, r), omega, omega_inv, pow(omega_inv, 6, r),  # l = blinding_factors + 1 = 6
         0, 0, 0, 0,          # has_accumulator, acc_offset, num_acc_limbs, num_acc_limb_bits
         1, 2,                # G1
         0, 0, 0, 0, 0, 0, 0, 0]  # G2, -s G2 (unused: pairing replaced, see module docstring)
    for pt in list(fixed_commitments) + list(permutation_commitments):
        w += [pt[0], pt[1]]
    assert len(w) * 32 == 0x03A0
    return w


def verify_proof(vk, proof, instances, srs_secret):
    cd = make_calldata(proof, instances)
    mem = {}

    def calldataload(o):
        return int.from_bytes(cd[o:o + 32].ljust(32, b"\0"), "big")

    def mload(o):
        return mem.get(o, 0)

    def mstore(o, v):
        mem[o] = v % (1 << 256)

    def membytes(start, length):
        out = bytearray()
        o = start
        while o < start + length:
            assert o % 32 == 0
            out += mload(o).to_bytes(32, "big")
            o += 32
        return bytes(out[:length])

    addmod = lambda a, b, m: (a + b) % m
    mulmod = lambda a, b, m: (a * b) % m
    sub = lambda a, b: (a - b) % (1 << 256)
    success = True

    def get_pt(o):
        x, y = mload(o), mload(o + 0x20)
        return None if x == 0 and y == 0 else (x, y)

    def put_pt(o, p):
        mstore(o, 0 if p is None else p[0])
        mstore(o + 0x20, 0 if p is None else p[1])

    # :77-87
    def read_ec_point(success, proof_cptr, hash_mptr):
        x, y = calldataload(proof_cptr), calldataload(proof_cptr + 0x20)
        ok = success and x < q and y < q and mulmod(y, y, q) == addmod(mulmod(x, mulmod(x, x, q), q), 3, q)
        mstore(hash_mptr, x)
        mstore(hash_mptr + 0x20, y)
        return ok, proof_cptr + 0x40, hash_mptr + 0x40

    # :93-99
    def squeeze_challenge(challenge_mptr, hash_mptr):
        h = int.from_bytes(keccak256(membytes(0x00, hash_mptr)), "big")
        mstore(challenge_mptr, h % r)
        mstore(0x00, h)
        return challenge_mptr + 0x20, 0x20

    # :106-112
    def squeeze_challenge_cont(challenge_mptr):
        h = int.from_bytes(keccak256(membytes(0x00, 0x20) + b"\x01"), "big")
        mstore(challenge_mptr, h % r)
        mstore(0x00, h)
        return challenge_mptr + 0x20

    # :116-159  (same values as the original's prefix-product + modexp dance)
    def batch_invert(success, mptr_start, mptr_end):
        o = mptr_start
        while o < mptr_end:
            v = mload(o)
            if v % r == 0:
                success = False
            else:
                mstore(o, pow(v, r - 2, r))
            o += 0x20
        return success

    def ec_add_acc(success, x, y):      # :163-167
        put_pt(0x00, P.g1_add(get_pt(0x00), None if x == 0 and y == 0 else (x, y)))
        return success

    def ec_mul_acc(success, scalar):    # :170-173
        put_pt(0x00, P.g1_mul(get_pt(0x00), scalar) if get_pt(0x00) else None)
        return success

    def ec_add_tmp(success, x, y):      # :177-181
        put_pt(0x80, P.g1_add(get_pt(0x80), None if x == 0 and y == 0 else (x, y)))
        return success

    def ec_mul_tmp(success, scalar):    # :185-188
        put_pt(0x80, P.g1_mul(get_pt(0x80), scalar) if get_pt(0x80) else None)
        return success

    # :216-305 — transcript
    for i, wv in enumerate(vk[:2]):                      # extcodecopy(vk, VK_MPTR, 0x00, 0x40)
        mstore(VK_MPTR + 0x20 * i, wv)
    success = success and calldataload(PROOF_LEN_CPTR) == 0x0460
    num_instances = mload(NUM_INSTANCES_MPTR)
    success = success and num_instances == calldataload(NUM_INSTANCE_CPTR)
    mstore(0x00, mload(VK_DIGEST_MPTR))
    hash_mptr = 0x20
    instance_cptr = INSTANCE_CPTR
    while instance_cptr < INSTANCE_CPTR + 0x20 * num_instances:
        inst = calldataload(instance_cptr)
        success = success and inst < r
        mstore(hash_mptr, inst)
        instance_cptr += 0x20
        hash_mptr += 0x20
    proof_cptr = PROOF_CPTR
    challenge_mptr = CHALLENGE_MPTR
    end = proof_cptr + 0x80                              # Phase 1 (:248-254)
    while proof_cptr < end:
        success, proof_cptr, hash_mptr = read_ec_point(success, proof_cptr, hash_mptr)
    challenge_mptr, hash_mptr = squeeze_challenge(challenge_mptr, hash_mptr)   # theta
    challenge_mptr = squeeze_challenge_cont(challenge_mptr)                    # beta
    challenge_mptr = squeeze_challenge_cont(challenge_mptr)                    # gamma
    end = proof_cptr + 0x0100                            # Phase 2 (:261-268)
    while proof_cptr < end:
        success, proof_cptr, hash_mptr = read_ec_point(success, proof_cptr, hash_mptr)
    challenge_mptr, hash_mptr = squeeze_challenge(challenge_mptr, hash_mptr)   # y
    end = proof_cptr + 0x80                              # Phase 3 (:272-279)
    while proof_cptr < end:
        success, proof_cptr, hash_mptr = read_ec_point(success, proof_cptr, hash_mptr)
    challenge_mptr, hash_mptr = squeeze_challenge(challenge_mptr, hash_mptr)   # x
    end = proof_cptr + 0x01E0                            # evaluations (:283-294)
    while proof_cptr < end:
        ev = calldataload(proof_cptr)
        success = success and ev < r
        mstore(hash_mptr, ev)
        proof_cptr += 0x20
        hash_mptr += 0x20
    challenge_mptr, hash_mptr = squeeze_challenge(challenge_mptr, hash_mptr)   # zeta (:297)
    challenge_mptr = squeeze_challenge_cont(challenge_mptr)                    # nu
    success, proof_cptr, hash_mptr = read_ec_point(success, proof_cptr, hash_mptr)  # W
    challenge_mptr, hash_mptr = squeeze_challenge(challenge_mptr, hash_mptr)   # mu
    success, proof_cptr, hash_mptr = read_ec_point(success, proof_cptr, hash_mptr)  # W'
    for i, wv in enumerate(vk):                          # extcodecopy(vk, VK_MPTR, 0x00, 0x03a0) (:307)
        mstore(VK_MPTR + 0x20 * i, wv)
    assert mload(HAS_ACCUMULATOR_MPTR) == 0              # :310-355 not exercised by this circuit

    # :358-435 — Lagrange evaluations and instance evaluation
    k = mload(K_MPTR)
    x = mload(X_MPTR)
    x_n = x
    for _ in range(k):
        x_n = mulmod(x_n, x_n, r)
    omega = mload(OMEGA_MPTR)
    mptr = X_N_MPTR
    mptr_end = mptr + 0x20 * (mload(NUM_INSTANCES_MPTR) + 6)
    if mload(NUM_INSTANCES_MPTR) == 0:
        mptr_end += 0x20
    pow_of_omega = mload(OMEGA_INV_TO_L_MPTR)
    while mptr < mptr_end:
        mstore(mptr, addmod(x, sub(r, pow_of_omega), r))
        pow_of_omega = mulmod(pow_of_omega, omega, r)
        mptr += 0x20
    x_n_minus_1 = addmod(x_n, sub(r, 1), r)
    mstore(mptr_end, x_n_minus_1)
    success = batch_invert(success, X_N_MPTR, mptr_end + 0x20)
    mptr = X_N_MPTR
    l_i_common = mulmod(x_n_minus_1, mload(N_INV_MPTR), r)
    pow_of_omega = mload(OMEGA_INV_TO_L_MPTR)
    while mptr < mptr_end:
        mstore(mptr, mulmod(l_i_common, mulmod(mload(mptr), pow_of_omega, r), r))
        pow_of_omega = mulmod(pow_of_omega, omega, r)
        mptr += 0x20
    l_blind = mload(X_N_MPTR + 0x20)
    l_i_cptr = X_N_MPTR + 0x40
    while l_i_cptr < X_N_MPTR + 0xC0:
        l_blind = addmod(l_blind, mload(l_i_cptr), r)
        l_i_cptr += 0x20
    instance_eval = 0
    instance_cptr = INSTANCE_CPTR
    while instance_cptr < INSTANCE_CPTR + 0x20 * mload(NUM_INSTANCES_MPTR):
        instance_eval = addmod(instance_eval, mulmod(mload(l_i_cptr), calldataload(instance_cptr), r), r)
        instance_cptr += 0x20
        l_i_cptr += 0x20
    x_n_minus_1_inv = mload(mptr_end)
    l_last = mload(X_N_MPTR)
    l_0 = mload(X_N_MPTR + 0xC0)
    mstore(X_N_MPTR, x_n)
    mstore(X_N_MINUS_1_INV_MPTR, x_n_minus_1_inv)
    mstore(L_LAST_MPTR, l_last)
    mstore(L_BLIND_MPTR, l_blind)
    mstore(L_0_MPTR, l_0)
    mstore(INSTANCE_EVAL_MPTR, instance_eval)

    # :438-512 — quotient evaluation
    delta = 4131629893567559867359510883348571134090853742863529169391034518566172092834
    y = mload(Y_MPTR)
    f_0, a_1, a_0 = calldataload(0x02C4), calldataload(0x02A4), calldataload(0x0284)
    numer = mulmod(f_0, addmod(a_1, sub(r, mulmod(a_0, a_0, r)), r), r)                                  # :443-450
    ev = addmod(mload(L_0_MPTR), sub(r, mulmod(mload(L_0_MPTR), calldataload(0x0364), r)), r)           # :452-456
    numer = addmod(mulmod(numer, y, r), ev, r)
    pz = calldataload(0x0424)                                                                             # :457-461
    ev = mulmod(mload(L_LAST_MPTR), addmod(mulmod(pz, pz, r), sub(r, pz), r), r)
    numer = addmod(mulmod(numer, y, r), ev, r)
    ev = mulmod(mload(L_0_MPTR), addmod(calldataload(0x03C4), sub(r, calldataload(0x03A4)), r), r)       # :462-465
    numer = addmod(mulmod(numer, y, r), ev, r)
    ev = mulmod(mload(L_0_MPTR), addmod(calldataload(0x0424), sub(r, calldataload(0x0404)), r), r)       # :466-469
    numer = addmod(mulmod(numer, y, r), ev, r)
    gamma, beta = mload(GAMMA_MPTR), mload(BETA_MPTR)
    lb = addmod(mload(L_LAST_MPTR), mload(L_BLIND_MPTR), r)
    # :470-482
    lhs = mulmod(calldataload(0x0384), addmod(addmod(calldataload(0x0284), mulmod(beta, calldataload(0x0304), r), r), gamma, r), r)
    mstore(0x00, mulmod(beta, mload(X_MPTR), r))
    rhs = mulmod(calldataload(0x0364), addmod(addmod(calldataload(0x0284), mload(0x00), r), gamma, r), r)
    mstore(0x00, mulmod(mload(0x00), delta, r))
    lsr = addmod(lhs, sub(r, rhs), r)
    numer = addmod(mulmod(numer, y, r), addmod(lsr, sub(r, mulmod(lsr, lb, r)), r), r)
    # :483-494
    lhs = mulmod(calldataload(0x03E4), addmod(addmod(calldataload(0x02A4), mulmod(beta, calldataload(0x0324), r), r), gamma, r), r)
    rhs = mulmod(calldataload(0x03C4), addmod(addmod(calldataload(0x02A4), mload(0x00), r), gamma, r), r)
    mstore(0x00, mulmod(mload(0x00), delta, r))
    lsr = addmod(lhs, sub(r, rhs), r)
    numer = addmod(mulmod(numer, y, r), addmod(lsr, sub(r, mulmod(lsr, lb, r)), r), r)
    # :495-505
    lhs = mulmod(calldataload(0x0444), addmod(addmod(mload(INSTANCE_EVAL_MPTR), mulmod(beta, calldataload(0x0344), r), r), gamma, r), r)
    rhs = mulmod(calldataload(0x0424), addmod(addmod(mload(INSTANCE_EVAL_MPTR), mload(0x00), r), gamma, r), r)
    lsr = addmod(lhs, sub(r, rhs), r)
    numer = addmod(mulmod(numer, y, r), addmod(lsr, sub(r, mulmod(lsr, lb, r)), r), r)
    mstore(QUOTIENT_EVAL_MPTR, mulmod(numer, mload(X_N_MINUS_1_INV_MPTR), r))                            # :510-511

    # :515-533 — quotient commitment
    mstore(0x00, calldataload(LAST_QUOTIENT_X_CPTR))
    mstore(0x20, calldataload(LAST_QUOTIENT_X_CPTR + 0x20))
    x_n = mload(X_N_MPTR)
    cptr, cptr_end = LAST_QUOTIENT_X_CPTR - 0x40, FIRST_QUOTIENT_X_CPTR - 0x40
    while cptr_end < cptr:
        success = ec_mul_acc(success, x_n)
        success = ec_add_acc(success, calldataload(cptr), calldataload(cptr + 0x20))
        cptr -= 0x40
    mstore(QUOTIENT_X_MPTR, mload(0x00))
    mstore(QUOTIENT_Y_MPTR, mload(0x20))

    # :536-781 — pairing lhs and rhs
    x = mload(X_MPTR)
    omega, omega_inv = mload(OMEGA_MPTR), mload(OMEGA_INV_MPTR)
    mstore(0x02C0, mulmod(x, omega, r))
    mstore(0x02A0, x)
    xp = mulmod(x, omega_inv, r)
    for _ in range(5):
        xp = mulmod(xp, omega_inv, r)
    mstore(0x0280, xp)
    mu = mload(MU_MPTR)                                                                                   # :553-580
    mptr, point_mptr = 0x02E0, 0x0280
    while mptr < 0x0340:
        mstore(mptr, addmod(mu, sub(r, mload(point_mptr)), r))
        mptr += 0x20
        point_mptr += 0x20
    mstore(0x0340, mload(0x0300))
    diff = mulmod(mload(0x02E0), mload(0x0320), r)
    mstore(0x0360, diff)
    mstore(0x00, diff)
    mstore(0x0380, 1)
    mstore(0x03A0, mload(0x02E0))
    mstore(0x20, mulmod(1, mload(0x0300), r))                                                             # :581-587
    p0, p1, p2 = mload(0x0280), mload(0x02A0), mload(0x02C0)                                              # :588-605
    mstore(0x40, mulmod(mulmod(addmod(p0, sub(r, p1), r), addmod(p0, sub(r, p2), r), r), mload(0x02E0), r))
    mstore(0x60, mulmod(mulmod(addmod(p1, sub(r, p0), r), addmod(p1, sub(r, p2), r), r), mload(0x0300), r))
    mstore(0x80, mulmod(mulmod(addmod(p2, sub(r, p0), r), addmod(p2, sub(r, p1), r), r), mload(0x0320), r))
    mstore(0xA0, mulmod(addmod(p1, sub(r, p2), r), mload(0x0300), r))                                     # :606-616
    mstore(0xC0, mulmod(addmod(p2, sub(r, p1), r), mload(0x0320), r))
    success = batch_invert(success, 0, 0xE0)                                                              # :617-631
    diff_0_inv = mload(0x00)
    mstore(0x0360, diff_0_inv)
    for mptr in (0x0380, 0x03A0):
        mstore(mptr, mulmod(mload(mptr), diff_0_inv, r))
    coeff, zeta = mload(0x20), mload(ZETA_MPTR)                                                           # :632-658
    r_eval = mulmod(coeff, calldataload(0x02E4), r)
    r_eval = mulmod(r_eval, zeta, r)
    r_eval = addmod(r_eval, mulmod(coeff, mload(QUOTIENT_EVAL_MPTR), r), r)
    cptr = 0x0344
    while 0x02E4 < cptr:
        r_eval = addmod(mulmod(r_eval, zeta, r), mulmod(coeff, calldataload(cptr), r), r)
        cptr -= 0x20
    cptr = 0x02C4
    while 0x0264 < cptr:
        r_eval = addmod(mulmod(r_eval, zeta, r), mulmod(coeff, calldataload(cptr), r), r)
        cptr -= 0x20
    mstore(0x03C0, r_eval)
    r_eval = 0                                                                                            # :659-671
    r_eval = addmod(r_eval, mulmod(mload(0x40), calldataload(0x0404), r), r)
    r_eval = addmod(r_eval, mulmod(mload(0x60), calldataload(0x03C4), r), r)
    r_eval = addmod(r_eval, mulmod(mload(0x80), calldataload(0x03E4), r), r)
    r_eval = mulmod(r_eval, zeta, r)
    r_eval = addmod(r_eval, mulmod(mload(0x40), calldataload(0x03A4), r), r)
    r_eval = addmod(r_eval, mulmod(mload(0x60), calldataload(0x0364), r), r)
    r_eval = addmod(r_eval, mulmod(mload(0x80), calldataload(0x0384), r), r)
    r_eval = mulmod(r_eval, mload(0x0380), r)
    mstore(0x03E0, r_eval)
    r_eval = 0                                                                                            # :672-679
    r_eval = addmod(r_eval, mulmod(mload(0xA0), calldataload(0x0424), r), r)
    r_eval = addmod(r_eval, mulmod(mload(0xC0), calldataload(0x0444), r), r)
    r_eval = mulmod(r_eval, mload(0x03A0), r)
    mstore(0x0400, r_eval)
    mstore(0x0420, mload(0x20))                                                                           # :680-694
    mstore(0x0440, addmod(addmod(mload(0x40), mload(0x60), r), mload(0x80), r))
    mstore(0x0460, addmod(mload(0xA0), mload(0xC0), r))
    for i in range(3):                                                                                    # :695-724
        mstore(0x20 * i, mload(0x0420 + 0x20 * i))
    success = batch_invert(success, 0, 0x60)
    r_eval = mulmod(mload(0x40), mload(0x0400), r)
    sum_inv_mptr, r_eval_mptr = 0x20, 0x03E0
    while sum_inv_mptr < 0x60:  # the original counts DOWN from 0x20 with unsigned wrap: visits 0x20, 0x00
        r_eval = mulmod(r_eval, mload(NU_MPTR), r)
        r_eval = addmod(r_eval, mulmod(mload(sum_inv_mptr), mload(r_eval_mptr), r), r)
        sum_inv_mptr = sub(sum_inv_mptr, 0x20)
        r_eval_mptr = sub(r_eval_mptr, 0x20)
    mstore(G1_SCALAR_MPTR, sub(r, r_eval))
    zeta, nu = mload(ZETA_MPTR), mload(NU_MPTR)                                                           # :725-772
    mstore(0x00, calldataload(0x01C4))
    mstore(0x20, calldataload(0x01E4))
    success = ec_mul_acc(success, zeta)
    success = ec_add_acc(success, mload(QUOTIENT_X_MPTR), mload(QUOTIENT_Y_MPTR))
    ptr = 0x07E0
    while 0x06E0 < ptr:
        success = ec_mul_acc(success, zeta)
        success = ec_add_acc(success, mload(ptr), mload(ptr + 0x20))
        ptr -= 0x40
    success = ec_mul_acc(success, zeta)
    success = ec_add_acc(success, calldataload(0xC4), calldataload(0xE4))
    success = ec_mul_acc(success, zeta)
    success = ec_add_acc(success, calldataload(0x84), calldataload(0xA4))
    mstore(0x80, calldataload(0x0144))
    mstore(0xA0, calldataload(0x0164))
    success = ec_mul_tmp(success, zeta)
    success = ec_add_tmp(success, calldataload(0x0104), calldataload(0x0124))
    success = ec_mul_tmp(success, mulmod(nu, mload(0x0380), r))
    success = ec_add_acc(success, mload(0x80), mload(0xA0))
    nu = mulmod(nu, mload(NU_MPTR), r)
    mstore(0x80, calldataload(0x0184))
    mstore(0xA0, calldataload(0x01A4))
    success = ec_mul_tmp(success, mulmod(nu, mload(0x03A0), r))
    success = ec_add_acc(success, mload(0x80), mload(0xA0))
    mstore(0x80, mload(G1_X_MPTR))
    mstore(0xA0, mload(G1_Y_MPTR))
    success = ec_mul_tmp(success, mload(G1_SCALAR_MPTR))
    success = ec_add_acc(success, mload(0x80), mload(0xA0))
    mstore(0x80, calldataload(0x0464))
    mstore(0xA0, calldataload(0x0484))
    success = ec_mul_tmp(success, sub(r, mload(0x0340)))
    success = ec_add_acc(success, mload(0x80), mload(0xA0))
    mstore(0x80, calldataload(0x04A4))
    mstore(0xA0, calldataload(0x04C4))
    success = ec_mul_tmp(success, mload(MU_MPTR))
    success = ec_add_acc(success, mload(0x80), mload(0xA0))
    mstore(PAIRING_LHS_X_MPTR, mload(0x00))
    mstore(PAIRING_LHS_Y_MPTR, mload(0x20))
    mstore(PAIRING_RHS_X_MPTR, calldataload(0x04A4))
    mstore(PAIRING_RHS_Y_MPTR, calldataload(0x04C4))

    # :811-817 — ec_pairing(lhs, rhs): e(lhs, G2) * e(rhs, -s G2) == 1  <=>  lhs == s * rhs
    lhs_pt, rhs_pt = get_pt(PAIRING_LHS_X_MPTR), get_pt(PAIRING_RHS_X_MPTR)
    success = success and lhs_pt == P.g1_mul(rhs_pt, srs_secret)
    return bool(success)
