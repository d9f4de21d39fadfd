"""Test helper: L1 C/A subframes that the reference's frame synchronisation accepts."""
import numpy as np

_L1CA_ROWS = ((0, 2, 3, 4, 6, 7, 11, 12, 13, 14, 15, 18, 19, 21, 24), (1, 3, 4, 5, 7, 8, 12, 13, 14, 15, 16, 19, 20, 22, 25),
              (0, 2, 4, 5, 6, 8, 9, 13, 14, 15, 16, 17, 20, 21, 23), (1, 3, 5, 6, 7, 9, 10, 14, 15, 16, 17, 18, 21, 22, 24),
              (1, 2, 4, 6, 7, 8, 10, 11, 15, 16, 17, 18, 19, 22, 23, 25), (0, 4, 6, 7, 9, 10, 11, 12, 14, 16, 20, 23, 24, 25))


def l1ca_subframe(rng, prev2, tow_count, sfid, polarity):
    """300 bits (+-1, as checkbit() would decide them) of one subframe that the reference's frame synchronisation
    accepts (ref src/sdrnav.c:325-346,373-411; src/sdrnav_gps.c:141-164), built by running its checks backwards:
    `b` is the frame as paritycheck() sees it after multiplying by the polarity; plain data bits, complemented for
    sending where the word before ended in -1, six parity bits from the products.  prev2: the two bits in front."""
    pre = [1, -1, -1, -1, 1, -1, 1, 1]
    b = list(prev2)
    for wd in range(10):
        d29, d30 = b[-2], b[-1]
        plain = [int(x) for x in rng.choice([-1, 1], size=24)]
        if wd == 0:
            plain[:8] = [v * (-1 if d30 == -1 else 1) for v in pre]          # so that the SENT bits are the preamble
        if wd == 1:
            # what decode_l1ca() will read: it works on the undecided-polarity bits r = polarity * b, complements them
            # back where r's word before ended in -1, and packs -1 as one
            want = [(tow_count >> (16 - i)) & 1 for i in range(17)] + [0, 0] + [(sfid >> (2 - i)) & 1 for i in range(3)]
            for i, bit in enumerate(want):
                r_after = -1 if bit else 1                                      # value after decode's complementing
                r_sent = r_after * (-1 if polarity * d30 == -1 else 1)
                b_sent = polarity * r_sent
                plain[i] = b_sent * (-1 if d30 == -1 else 1)
        w = [d29, d30] + plain
        par = []
        for row in _L1CA_ROWS:
            prod = 1
            for k in row:
                prod *= w[k]
            par.append(prod)
        sent = [v * (-1 if d30 == -1 else 1) for v in plain] + par
        b += sent
    return [polarity * v for v in b[2:]], b[-2:]
