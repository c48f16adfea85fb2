#!/usr/bin/env python3
"""Developer tool: the hops of a resident back-substitution from its stamps (SK_BS_STAMPS=<file>, whole-system launch): per block column
the owner's wait for the block row next to the diagonal, the hand-over (producer's store -> consumer past its barrier) and the
consumer's own part (-> its store)."""
import sys
import numpy as np
a = np.loadtxt(sys.argv[1])  # rows: kb descending; columns: kb, wait start, arrival, stored
stored, arrival, wait0 = a[:, 3], a[:, 2], a[:, 1]
hop = np.diff(stored) * 0.01
hand = (arrival[1:] - stored[:-1]) * 0.01
own = (stored[1:] - arrival[1:]) * 0.01
idle = (arrival - wait0)[1:] * 0.01
print("%s: %d hops, median %.2f us = hand-over %.2f + consumer's part %.2f; the consumer had been waiting for %.2f us" % (sys.argv[1], len(hop), np.median(hop), np.median(hand), np.median(own), np.median(idle)))
