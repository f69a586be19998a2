import sys, os
sys.path.insert(0, os.getcwd()); import ltompc, torch
T = ltompc.build_tables()
mode = sys.argv[1]
if mode == "lib_first":
    m = ltompc.BatchedMPC(T, 10, 4)
    print("handle ok")
    x = torch.zeros(4, device="cuda:0"); print("torch ok", x.sum().item())
elif mode == "split_first":
    m = ltompc.BatchedMPC(T, 10, 4)
    s = ltompc.SplitMPC(T, 10, 8, n_parts=2)
    print("handles ok")
    x = torch.zeros(4, device="cuda:0"); print("torch ok", x.sum().item())
else:
    x = torch.zeros(4, device="cuda:0"); print("torch ok")
    m = ltompc.BatchedMPC(T, 10, 4); print("handle ok")
