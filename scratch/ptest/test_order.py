import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
def test_order():
    import ltompc
    T = ltompc.build_tables()
    m = ltompc.BatchedMPC(T, 10, 4)
    import torch
    x = torch.zeros(4, device="cuda:0")
    assert float(x.sum()) == 0.0
