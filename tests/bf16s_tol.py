"""Config 5 (precision 'bf16s': bf16 arithmetic, bf16 activations and weight copies, fp32 masters / gradients / slots): the
tolerance each tensor class is held to against the fp32 oracle = 1.5 x the worst rel-L2 observed for the class at B = 64
(tools/bf16s_errors.py, profiles/r05_bf16s_errors.txt; DESIGN 0 holds the table).  Depth maps against the oracle's forward;
gradients against the oracle's fp32 backward of the activations the replica stored."""

DEPTH = {'coarse': 8.5e-3, 'fine': 4.0e-3}            # observed 5.6e-3 / 2.7e-3


def grad_tol(name):
    if name.startswith('coarse/conv/'):
        return 1.1e-2                                   # observed <= 6.8e-3 (conv2d_0: the longest chain of bf16 gradients)
    if name.startswith('coarse/dense/dense_0'):
        return 5.3e-3                                   # observed 3.5e-3 (bf16 c4 and dz0 enter; fp32 accumulate)
    if name.startswith('coarse/dense/dense_1') or name.startswith('fine/third'):
        return 1e-5                                     # fp32 operands in this mode too: observed 3e-8 / 6e-7
    if name.startswith('fine/second'):
        return 3.2e-3                                   # observed 2.1e-3
    if name.startswith('fine/first'):
        return 5.5e-3                                   # observed 3.6e-3 (its filter gradient in bf16 arithmetic: fewch16.hip)
    raise KeyError(name)
