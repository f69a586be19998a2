import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
def test_torch_first():
    import torch
    x = torch.zeros(4, device="cuda:0")
    import ltompc
    T = ltompc.build_tables()
    m = ltompc.BatchedMPC(T, 10, 4)
    assert float(x.sum()) == 0.0
