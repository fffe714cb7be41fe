"""CPU suite: the Fiat-Shamir transcript mirror.  Keccak-f[1600] is pinned against
hashlib's SHA3, the Merlin/STROBE layer against the merlin crate's published test transcript, the G1
encoding against the standard compressed BLS12-381 generator."""
import hashlib


def _sha3_256(T, msg: bytes) -> bytes:
    rate = 136
    st = bytearray(200)
    m = bytearray(msg) + b"\x06"
    while len(m) % rate:
        m += b"\x00"
    m[-1] |= 0x80
    for off in range(0, len(m), rate):
        for i in range(rate):
            st[i] ^= m[off + i]
        T.keccak_f1600(st)
    return bytes(st[:32])


def test_keccak_f1600_against_hashlib(mj):
    T = mj.transcript
    st1, st2 = bytearray(range(200)), bytearray(range(200))
    T.keccak_f1600(st1)                                   # library host function
    T.keccak_f1600_py(st2)                                # pure-Python cross-check
    assert st1 == st2
    for msg in (b"", b"abc", b"\x00" * 135, b"\xff" * 136, b"jellyfish" * 50):
        assert _sha3_256(T, msg) == hashlib.sha3_256(msg).digest()


def test_merlin_known_answer(mj):
    """merlin's own test transcript: Transcript::new(b"test protocol"); append_message(b"some label", b"some data");
    challenge_bytes(b"challenge", 32)."""
    t = mj.transcript.MerlinTranscript(b"test protocol")
    t.append_message(b"some label", b"some data")
    assert t.challenge_bytes(b"challenge", 32).hex() == "d5a21972d0d5fe320c0d263fac7fffb8145aa640af6e9bca177c03c7efcf0615"
    # a message crossing the STROBE rate (166 bytes) exercises run_f inside absorb and squeeze
    t.append_message(b"big", bytes(range(256)) * 3)
    a = t.challenge_bytes(b"c2", 400)
    t2 = mj.transcript.MerlinTranscript(b"test protocol")
    t2.append_message(b"some label", b"some data")
    t2.challenge_bytes(b"challenge", 32)
    t2.append_message(b"big", bytes(range(256)) * 3)
    assert t2.challenge_bytes(b"c2", 400) == a and len(set(a)) > 100


def test_encodings(mj):
    T, P = mj.transcript, mj.params
    bls, bn = P.BLS12_381, P.BN254
    # the well-known compressed BLS12-381 G1 generator (IETF / Zcash form)
    assert T.g1_bytes(bls, (bls.gx, bls.gy)).hex() == (
        "97f1d3a73197d7942695638c4fa9ac0fc3688c4f9774b905a14e3a3f171bac586c55e83ff97a1aeffb3af00adb22c6bb")
    neg = (bls.gx, bls.q - bls.gy)
    assert T.g1_bytes(bls, neg)[0] == 0x97 | 0x20 and T.g1_bytes(bls, neg)[1:] == T.g1_bytes(bls, (bls.gx, bls.gy))[1:]
    assert T.g1_bytes(bls, None) == bytes([0xC0]) + bytes(47)
    assert T.g1_bytes(bn, (1, 2)) == (1).to_bytes(32, "little")                      # y = 2 < q/2: no flag
    assert T.g1_bytes(bn, (1, bn.q - 2))[31] == 0x80 and T.g1_bytes(bn, None)[31] == 0x40
    assert T.fr_bytes(bls, bls.r + 5) == (5).to_bytes(32, "little")


def test_standard_transcript_challenges(mj):
    """64 squeezed bytes reduced mod r, re-absorbed (standard.rs:33-45): the next challenge depends on the previous."""
    c = mj.params.BLS12_381
    t = mj.transcript.StandardTranscript(c)
    t.append_vk_and_pub_input(1 << 10, 1, [1, 2, 3, 4, 5], [(c.gx, c.gy)] * 13, [None] * 5, [42])
    t.append_commitments(b"witness_poly_comms", [(c.gx, c.gy)] * 5)
    tau, beta, gamma = (t.get_and_append_challenge(x) for x in (b"tau", b"beta", b"gamma"))
    assert len({tau, beta, gamma}) == 3 and all(0 <= x < c.r for x in (tau, beta, gamma))
    u = mj.transcript.StandardTranscript(c)
    u.append_vk_and_pub_input(1 << 10, 1, [1, 2, 3, 4, 5], [(c.gx, c.gy)] * 13, [None] * 5, [42])
    u.append_commitments(b"witness_poly_comms", [(c.gx, c.gy)] * 5)
    assert u.get_and_append_challenge(b"tau") == tau
    u2 = mj.transcript.StandardTranscript(c)
    u2.append_vk_and_pub_input(1 << 10, 1, [1, 2, 3, 4, 5], [(c.gx, c.gy)] * 13, [None] * 5, [43])   # different public input
    u2.append_commitments(b"witness_poly_comms", [(c.gx, c.gy)] * 5)
    assert u2.get_and_append_challenge(b"tau") != tau


def test_chacha_and_field_sampling(mj):
    """rand_chacha: ChaCha20 / ChaCha12 / ChaCha8 keystream blocks for the all-zero key (RFC 7539 2.3.2-style vector and the
    ChaCha test-vector draft TC1); BlockRng's u64 = two consecutive words; ark-ff's rejection sampling stays below r."""
    import struct
    R = mj.rng
    blk = lambda rounds: struct.pack("<16I", *R.chacha_block((0,) * 8, 0, rounds)).hex()
    assert blk(20) == ("76b8e0ada0f13d90405d6ae55386bd28bdd219b8a08ded1aa836efcc8b770dc7"
                       "da41597c5157488d7724e03fb8d84a376a43b8f41518a11cc387b669b2ee6586")
    assert blk(12) == ("9bf49a6a0755f953811fce125f2683d50429c3bb49e074147e0089a52eae155f"
                       "0564f879d27ae3c02ce82834acfa8c793a629f2ca0de6919610be82f411326be")
    assert blk(8) == ("3e00ef2f895f40d67f5bb8e81f09a5a12c840ec3ce9a7f3b181be188ef711a1e"
                      "984ce172b9216f419f445367456d5619314a42a3da86b001387bfdb80e0cfe42")
    g = R.ChaChaRng(bytes(32), 20)
    words = R.chacha_block((0,) * 8, 0, 20)
    assert g.next_u64() == words[0] | (words[1] << 32) and g.next_u32() == words[2]
    g2 = R.ChaChaRng(bytes(32), 20)
    for _ in range(32):
        g2.next_u64()                                         # drains the four-block buffer exactly
    assert g2.next_u64() == (lambda w: w[0] | (w[1] << 32))(R.chacha_block((0,) * 8, 4, 20))
    for cid in (0, 1):
        c = mj.params.CURVES[cid]
        rng = R.test_rng()
        vals = [R.fr_rand(c, rng) for _ in range(50)]
        assert all(0 <= v < c.r for v in vals) and len(set(vals)) == 50
        ks = R.compute_coset_representatives(c, 6, 1 << 10)
        assert ks[0] == 1 and len({pow(k, 1 << 10, c.r) for k in ks}) == 6           # six distinct cosets of H
        assert R.compute_coset_representatives(c, 5, 1 << 10) == ks[:5]
