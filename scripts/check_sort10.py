"""0/1-principle check of the 10-input sorting network of kernels_init.hip (sweep_sort10): every 0/1 input comes out sorted, 29 compare-exchanges."""
import itertools
NET = [[(0, 8), (1, 9), (2, 7), (3, 5), (4, 6)], [(0, 2), (1, 4), (5, 8), (7, 9)], [(0, 3), (2, 4), (5, 7), (6, 9)], [(0, 1), (3, 6), (8, 9)],
       [(1, 5), (2, 3), (4, 8), (6, 7)], [(1, 2), (3, 5), (4, 6), (7, 8)], [(2, 3), (4, 5), (6, 7)], [(3, 4), (5, 6)]]
for bits in itertools.product([0, 1], repeat=10):
    v = list(bits)
    for layer in NET:
        for a, b in layer:
            if v[a] > v[b]:
                v[a], v[b] = v[b], v[a]
    assert v == sorted(v), bits
print("ok:", sum(len(l) for l in NET), "compare-exchanges")
