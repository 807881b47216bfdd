/* mcr_sortnet.h -- sorting networks for the per-lane register sort: 16 inputs (60 compare-exchanges in 10 layers)
 * and 8 inputs (19 in 6); the 16-input bitonic merge of the register merge levels.
 * Plain C so that the host-side test (tests/test_host_cpu.py) can verify it exhaustively with the
 * 0-1 principle (all 65536 binary inputs).  X(a, b) = compare-exchange positions a < b. */
#ifndef MCR_SORTNET_H
#define MCR_SORTNET_H
#define MCR_NET16(X)                                                                       \
    X(0, 13) X(1, 12) X(2, 15) X(3, 14) X(4, 8) X(5, 6) X(7, 11) X(9, 10)                 \
    X(0, 5) X(1, 7) X(2, 9) X(3, 4) X(6, 13) X(8, 14) X(10, 15) X(11, 12)                 \
    X(0, 1) X(2, 3) X(4, 5) X(6, 8) X(7, 9) X(10, 11) X(12, 13) X(14, 15)                 \
    X(0, 2) X(1, 3) X(4, 10) X(5, 11) X(6, 7) X(8, 9) X(12, 14) X(13, 15)                 \
    X(1, 2) X(3, 12) X(4, 6) X(5, 7) X(8, 10) X(9, 11) X(13, 14)                          \
    X(1, 4) X(2, 6) X(5, 8) X(7, 10) X(9, 13) X(11, 14)                                   \
    X(2, 4) X(3, 6) X(9, 12) X(11, 13)                                                    \
    X(3, 5) X(6, 8) X(7, 9) X(10, 12)                                                     \
    X(3, 4) X(5, 6) X(7, 8) X(9, 10) X(11, 12)                                            \
    X(6, 7) X(8, 9)
/* 8-input network, 19 compare-exchanges in 6 layers (the tile sort's 512-thread x 8-draw configuration). */
#define MCR_NET8(X)                                                                        \
    X(0, 2) X(1, 3) X(4, 6) X(5, 7)                                                        \
    X(0, 4) X(1, 5) X(2, 6) X(3, 7)                                                        \
    X(0, 1) X(2, 3) X(4, 5) X(6, 7)                                                        \
    X(2, 4) X(3, 5)                                                                        \
    X(1, 4) X(3, 6)                                                                        \
    X(1, 2) X(3, 4) X(5, 6)
/* Bitonic merge of 16 inputs (a bitonic sequence in, ascending out): half-cleaners at distance 8, 4, 2, 1.  The local
 * step of the tile sort's register merge levels (lane_bitonic_merge16); tests/test_host_cpu.py replays those levels --
 * mirror exchange with the partner lane, half-cleaners across lanes, this network inside the lane -- on every pair of
 * sorted 0-1 runs. */
#define MCR_BITONIC16(X)                                                                   \
    X(0, 8) X(1, 9) X(2, 10) X(3, 11) X(4, 12) X(5, 13) X(6, 14) X(7, 15)                 \
    X(0, 4) X(1, 5) X(2, 6) X(3, 7) X(8, 12) X(9, 13) X(10, 14) X(11, 15)                 \
    X(0, 2) X(1, 3) X(4, 6) X(5, 7) X(8, 10) X(9, 11) X(12, 14) X(13, 15)                 \
    X(0, 1) X(2, 3) X(4, 5) X(6, 7) X(8, 9) X(10, 11) X(12, 13) X(14, 15)
#endif
