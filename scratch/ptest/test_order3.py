def test_torch_only():
    import torch
    x = torch.zeros(4, device="cuda:0")
    assert float(x.sum()) == 0.0
