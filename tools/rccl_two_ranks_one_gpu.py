"""Try the multi-GPU exchange with TWO ranks on ONE GPU (both processes use device 0).  RCCL normally refuses duplicate devices;
this probes whether this build allows it (it lets the in-place reduce-scatter / grouped broadcasts / deferred all-gather run
for real at world size 2 on a 1-GPU box).  python tools/rccl_two_ranks_one_gpu.py"""
import os, sys, socket
import multiprocessing as mp
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK='0', MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port),
                      HSA_ENABLE_IPC_MODE_LEGACY='0')
    import numpy as np
    import adorym_amd as A
    from adorym_amd import comm as C
    from adorym_amd.dp import DataParallelObject, HipOps
    try:
        ctx = A.Context(0)
        rc = C.RcclComm(device_index=0).attach(ctx)
        shape = (8, 5, 6, 2)
        n = int(np.prod(shape))
        st = DataParallelObject(HipOps(ctx), rc, shape)
        r = np.random.default_rng(7)
        st.obj.view(0, (n,)).set((r.standard_normal(n) * 1e-3).astype(np.float32))
        for it in range(3):
            st.zero_grad()
            st.grad.view(0, (n,)).set(np.random.default_rng(100 * it + rank).standard_normal(n).astype(np.float32))
            st.exchange_and_update('adam', it, {'step_size': 1e-4}, flags=1, first=(100, 300))
            ctx.fork(); st.finish_update(); ctx.end_fork(); ctx.join()
        out = st.obj.view(0, (n,)).get()
        rc.barrier()
        q.put((rank, 'ok', out))
        rc.close()
    except Exception as e:
        q.put((rank, 'error: %r' % (e,), None))


if __name__ == '__main__':
    s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    ps = [ctx.Process(target=worker, args=(r, 2, port, q)) for r in range(2)]
    [p.start() for p in ps]
    res = []
    try:
        res = [q.get(timeout=120) for _ in ps]
    except Exception as e:
        print('timeout / no answer:', repr(e))
    for p in ps:
        p.join(10)
        if p.is_alive():
            p.terminate()
    for r in sorted(res, key=lambda t: t[0]):
        print('rank', r[0], r[1])
    if len(res) == 2 and all(r[1] == 'ok' for r in res):
        import numpy as np
        sys.path.insert(0, ROOT)
        from oracle import adorym_oracle as O       # checker only
        n = res[0][2].size
        x = (np.random.default_rng(7).standard_normal(n) * 1e-3).astype(np.float32)
        m = np.zeros(n, np.float32); v = np.zeros(n, np.float32)
        for it in range(3):
            g = sum(np.random.default_rng(100 * it + k).standard_normal(n).astype(np.float32) for k in range(2))
            x, m, v = O.adam_step(x, g, m, v, it, 1e-4)
            x = np.clip(x, 0, None)
        print('ranks agree:', np.array_equal(res[0][2], res[1][2]), ' vs oracle max rel err: %.2e' % (np.abs(res[0][2] - x).max() / np.abs(x).max()))
