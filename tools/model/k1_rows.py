"""Semantic model (numpy, one 64-lane vector per wave value) of the row-parallel LZ4 decode kernel
(4mc_amd/csrc/lz4_rows.hip).  Development aid: it checks the arithmetic of the four roles (row tables by pointer
doubling, the serial walk, token records + literal scatter, match-space copier) against the bytes the stream decodes to;
it does not model the concurrency.  Usage: python tools/model/k1_rows.py [block ...]"""
import sys
import numpy as np

ROW = 64
LANE = np.arange(64, dtype=np.int64)
OWN = 2048
NQ = 8


def u32_at(buf, pos):            # little-endian dword at arbitrary byte positions (buf is padded)
    return (buf[pos].astype(np.int64) | (buf[pos + 1].astype(np.int64) << 8) |
            (buf[pos + 2].astype(np.int64) << 16) | (buf[pos + 3].astype(np.int64) << 24))


def pre_row(buf, r):
    """stage 1-3: per-lane token speculation + doubling.  Returns (tab[64], fields dict)."""
    pos = r * ROW + LANE
    w = u32_at(buf, pos)
    b = w & 0xff; b1 = (w >> 8) & 0xff
    L0 = b >> 4; M0 = b & 15
    lext = (L0 == 15)
    L = L0 + np.where(lext, b1, 0)
    offpos = LANE + 1 + lext + L
    wo = u32_at(buf, r * ROW + offpos)
    off = wo & 0xffff; e1 = (wo >> 16) & 0xff
    mext = (M0 == 15)
    ml = M0 + 4 + np.where(mext, e1, 0)
    nxt = offpos + 2 + mext
    reg = ~(lext & (b1 == 255)) & ~(mext & (e1 == 255))
    nxt = np.where(reg, nxt, LANE)
    pz = np.where(reg, (L << 16) | ml, 0)
    nonterm = reg & (nxt < 64)
    h = np.where(nonterm, nxt, LANE)
    h0 = h.copy()
    s = np.where(nonterm, pz, 0)
    for _ in range(5):
        s = s + s[h]
        h = h[h]
    # 21 tokens at most in 64 bytes: 5 rounds reach the terminal
    assert (h[h] == h).all()
    tot = s + pz[h]
    X = nxt[h]
    lits = tot >> 16; mls = tot & 0xffff
    over = (lits > 2047) | (mls > 4095)
    X = np.where(over, LANE, X); lits = np.where(over, 0, lits); mls = np.where(over, 0, mls)
    tab = (X << 23) | (lits << 12) | mls
    f = dict(off=off, L=L, ml=ml, lext=lext.astype(np.int64), reg=reg, h0=h0, byte=b, pz=pz, over=over)
    return tab, f


def scan_add(v):
    return np.cumsum(v)


def scan_max(v):
    return np.maximum.accumulate(v)


class Fail(Exception):
    pass


def decode_block(src, cap):
    """returns decoded bytes (np.uint8) or raises Fail (-> the exact kernel would redo the block)."""
    csize = len(src)
    buf = np.concatenate([src, np.zeros(1024, np.uint8)])
    out = np.zeros(cap + 64, np.uint8)
    iend, oend = csize, cap
    nrows = (csize + 63) // 64
    rlast = (iend - 16 - 337 - 64) // 64      # last row whose regular tokens all end 16 bytes before iend
    tabs = {}; flds = {}
    def table(r):
        if r not in tabs: tabs[r], flds[r] = pre_row(buf, r)
        return tabs[r]
    own = np.zeros(OWN, np.int64)
    recs = {}              # q -> dict(off, ovl, D, msd, mend)
    stats = dict(rows=0, general=0, steps=0, passes=0)
    # ---------------- W: the walk
    p = 0; op = 0; mb = 0; q = 0
    visited = []           # (r, e, op, mb, q)
    general = []           # (q or None, op, lit, lit_ip, off, mlen, mb)
    row_general = -1       # row in which W stays in general mode
    end_value = None
    while True:
        r = p >> 6; e = p & 63
        batch = False
        if r <= rlast and r != row_general:
            t = int(table(r)[e])
            X = t >> 23; lits = (t >> 12) & 0x7ff; mls = t & 0xfff
            if X != e and op + lits + mls + 80 <= oend:
                batch = True
        if batch:
            visited.append((r, e, op, mb, q))
            op += lits + mls; mb += mls; p = r * 64 + X; q += 1
            stats['rows'] += 1
            continue
        # ---- one general sequence, strict rules (lz4.c safe loop)
        stats['general'] += 1
        row_general = r
        ip = p
        if ip >= iend: raise Fail("ip>=iend")
        token = int(buf[ip]); ip += 1
        lit = token >> 4; mlen = token & 15
        if lit == 15:
            if ip >= iend - 15: raise Fail("litlen")
            while True:
                c = int(buf[ip]); ip += 1; lit += c
                if ip > iend - 15 and c == 255: raise Fail("litlen2")   # model only: conservative
                if c != 255: break
        if op + lit > oend - 12 or ip + lit > iend - 8:
            if ip + lit != iend or op + lit > oend: raise Fail("tail")
            general.append((None, op, lit, ip, 0, 0, mb))
            end_value = op + lit
            break
        lit_ip = ip; ip += lit; op2 = op + lit
        off = int(buf[ip]) | (int(buf[ip + 1]) << 8); ip += 2
        if mlen == 15:
            while True:
                if ip >= iend - 4: raise Fail("mlen")
                c = int(buf[ip]); ip += 1; mlen += c
                if c != 255: break
        mlen += 4
        if off == 0 or off > op2 or op2 + mlen > oend - 5: raise Fail("match")
        general.append((q, op, lit, lit_ip, off, mlen, mb))
        op = op2 + mlen; mb += mlen; p = ip; q += 1
    total_q = q
    # ---------------- general sequences: literals by W, match record
    for (gq, gop, lit, lit_ip, off, mlen, gmb) in general:
        out[gop:gop + lit] = buf[lit_ip:lit_ip + lit]
        if gq is not None:
            own[gmb & (OWN - 1)] = -1          # model: own entries are filled lazily below (ring not modelled)
            recs[gq] = dict(lane_off={0: off}, lane_D={0: (gop + lit) - gmb}, lane_msd={0: gop + lit}, mbeg=gmb, mend=gmb + mlen,
                            starts={gmb: 0})
    # ---------------- post: records + literal scatter, row by row with the literal carry
    vis = {r: (e, vop, vmb, vq) for (r, e, vop, vmb, vq) in visited}
    carry = None           # (litlo_rel_to_row, L, dest_of_lane0)
    for r in range(0, nrows):
        if r not in vis and carry is None: continue
        key = np.zeros(64, np.int64)
        base = None
        if r in vis:
            e, vop, vmb, vq = vis[r]
            f = flds[r]
            # marking: walk the level-0 hops from e
            mark = np.zeros(64, bool); j = e
            while True:
                mark[j] = True
                if f['h0'][j] == j: break
                j = int(f['h0'][j])
            is_tok = mark & f['reg']
            pz = np.where(is_tok, f['pz'], 0)
            incl = scan_add(pz); excl = incl - pz
            litcum_incl = incl >> 16; mlcum_excl = excl & 0xffff; litcum_excl = excl >> 16
            D = (vop - vmb) + litcum_incl
            mpos = vmb + mlcum_excl
            msd = mpos + D
            bad = is_tok & ((f['off'] == 0) | (f['off'] > msd))
            if bad.any(): raise Fail("offset")
            tl = np.nonzero(is_tok)[0]
            recs[vq] = dict(lane_off={int(l): int(f['off'][l]) for l in tl}, lane_D={int(l): int(D[l]) for l in tl},
                            lane_msd={int(l): int(msd[l]) for l in tl}, mbeg=vmb, mend=vmb + int(incl[63] & 0xffff),
                            starts={int(mpos[l]): int(l) for l in tl})
            litlo = LANE + 1 + f['lext']
            os_i = litcum_excl + mlcum_excl
            base = vop - 512
            dl = os_i - litlo + 512
            assert (dl[is_tok] >= 0).all() and (dl[is_tok] < 16384).all()
            key = np.where(is_tok, ((litlo + 320) << 23) | (f['L'] << 14) | dl, 0)
        if carry is not None:
            clitlo, cL, cdest0 = carry
            if base is None: base = cdest0 - 512
            dl0 = cdest0 - base
            assert 0 <= dl0 < 16384 and clitlo + 320 >= 0, (dl0, clitlo)
            if key[0] == 0: key[0] = ((clitlo + 320) << 23) | (cL << 14) | dl0
            else: assert clitlo + cL <= 0          # a token at lane 0: the carried run has ended
        km = scan_max(key)
        litlo_k = (km >> 23) - 320
        rel = LANE - litlo_k
        Lk = (km >> 14) & 0x1ff
        is_lit = (km != 0) & (rel >= 0) & (rel < Lk)
        # carried key: dest = lane + (dl0) + base;  token key: dest = lane + dl + base  (dl is "dest of lane 0" in both)
        dest = LANE + (km & 0x3fff) + base
        bytes_row = buf[r * 64 + LANE]
        out[dest[is_lit]] = bytes_row[is_lit]
        # next carry: from the last key of the row
        k63 = int(km[63])
        if k63 == 0: carry = None
        else:
            nlitlo = (k63 >> 23) - 320 - 64; nL = (k63 >> 14) & 0x1ff
            ndest0 = 64 + (k63 & 0x3fff) + base
            carry = (nlitlo, nL, ndest0) if nlitlo + nL > 0 else None
    # ---------------- C: match-space copier with multi-pass dependency handling
    qorder = sorted(recs)
    assert qorder == list(range(total_q)), (len(qorder), total_q)
    ext = recs[total_q - 1]['mend'] if total_q else 0
    # owner lookup structure (stands for own ring + scan_max with carry)
    starts = []
    for qq in qorder:
        for mp, l in sorted(recs[qq]['starts'].items()): starts.append((mp, qq, l))
    sm = np.array([s[0] for s in starts], np.int64)
    g = 0
    while g < ext:
        m = g + LANE
        live = m < ext
        idx = np.searchsorted(sm, np.minimum(m, ext - 1), side='right') - 1
        off = np.array([recs[starts[i][1]]['lane_off'][starts[i][2]] for i in idx], np.int64)
        D = np.array([recs[starts[i][1]]['lane_D'][starts[i][2]] for i in idx], np.int64)
        msd = np.array([recs[starts[i][1]]['lane_msd'][starts[i][2]] for i in idx], np.int64)
        dest = m + D
        rel = dest - msd
        src = dest - off
        ov = rel >= off
        src = np.where(ov, msd - off + rel % np.maximum(off, 1), src)
        done = ~live
        stats['steps'] += 1
        while not done.all():
            first = int(np.argmin(done))
            bound = dest[first]
            ready = ~done & (src < bound)
            assert ready[first]
            out[dest[ready]] = out[src[ready]]
            done |= ready
            stats['passes'] += 1
        g += 64
    return out[:end_value], stats


def main():
    sys.path.insert(0, __file__.rsplit('/tools/', 1)[0] + '/tests')
    import helpers as H
    O = H.oracle()
    B = 4 << 20
    blocks = [int(a) for a in sys.argv[1:]] or [0, 1, 2, 3, 5]
    n = int(__import__('os').environ.get('MODEL_BYTES', str(256 << 10)))
    for b in blocks:
        data = H.corpus(B, first_block=b)[:n].copy()
        dst = np.empty(n + n // 255 + 64, np.uint8)
        cs = O.orc_lz4_compress_fast(data.ctypes.data, dst.ctypes.data, n, n - 1)
        if cs <= 0: print(b, "stored"); continue
        try:
            out, st = decode_block(dst[:cs].copy(), n)
        except Fail as e:
            print(b, "FAIL ->retry", e); continue
        ok = len(out) == n and (out == data).all()
        print(f"block {b}: csize {cs} ok={ok} {st}  passes/step {st['passes']/max(st['steps'],1):.2f}")
        if not ok:
            bad = np.nonzero(out[:min(len(out), n)] != data[:min(len(out), n)])[0]
            print("  first mismatch at", bad[:5], "len", len(out))


if __name__ == '__main__':
    main()
