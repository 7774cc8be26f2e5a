import sys, time, torch
sys.path.insert(0, ".")
from spacefortress_amd import SFVecEnv
for gt, na in (("youturn", 5), ("autoturn", 3)):
    for K in (16, 64, 256):
        n = 65536
        env = SFVecEnv(n, gametype=gt, spawn_stride=1)
        acts = torch.randint(0, na, (K, n), device=env.device, dtype=torch.uint8)
        obs = torch.empty((K, n, env.obs_dim), device=env.device)
        rew = torch.empty((K, n), dtype=torch.int32, device=env.device)
        done = torch.empty((K, n), dtype=torch.uint8, device=env.device)
        info = torch.empty((K, n), dtype=torch.uint8, device=env.device)
        for _ in range(3):
            env.rollout(acts, out=(obs, rew, done, info))
        torch.cuda.synchronize()
        L = max(4, 2048 // K)
        t0 = time.perf_counter()
        for _ in range(L):
            env.rollout(acts, out=(obs, rew, done, info))
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print(gt, "K=%d: %.2f us/step, %.2f Gsteps/s" % (K, dt / (L * K) * 1e6, n * L * K / dt / 1e9))
        env.close()
