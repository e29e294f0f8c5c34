"""What of the RCCL path can run on a ONE-GPU box (no multi-GPU node was available in any round).

  python tools/diag/rccl_probe.py            # prints a small report; exit code 0 whatever RCCL answers

  leg 1  world 1, backend nccl (= RCCL): process group on cuda:0, `sharding.warm_up` (the collective every entry point
         of sharding.py starts with), all_reduce / all_gather_object / barrier, a hand-off message packed and checked on
         the device -- everything of the multi-GPU path except the point-to-point call itself.
  leg 2  world 2, BOTH ranks on cuda:0: the header-checked isend / recv of `sharding.chained_predict`.  RCCL (like NCCL)
         is expected to refuse a communicator with two ranks on one device; the leg records what it says.  Each rank runs
         in its own process under a deadline and is killed by PID if it hangs.
"""
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def free_port():
  s = socket.socket()
  s.bind(('127.0.0.1', 0))
  p = s.getsockname()[1]
  s.close()
  return p


def rank_main(rank, world, port):
  import datetime
  import torch
  import torch.distributed as dist
  from msd_amd import sharding
  os.environ['MASTER_ADDR'] = '127.0.0.1'
  os.environ['MASTER_PORT'] = str(port)
  torch.cuda.set_device(0)
  dev = torch.device('cuda', 0)
  t0 = time.perf_counter()
  dist.init_process_group('nccl', rank=rank, world_size=world, device_id=dev, timeout=datetime.timedelta(seconds=60))
  print('[rank %d/%d] process group up in %.2f s (torch %s, nccl/rccl %s)'
        % (rank, world, time.perf_counter() - t0, torch.__version__, '.'.join(map(str, torch.cuda.nccl.version()))), flush=True)
  t0 = time.perf_counter()
  sharding.warm_up(None, dev)
  print('[rank %d] sharding.warm_up (first collective, communicator setup): %.2f s' % (rank, time.perf_counter() - t0), flush=True)
  x = torch.full((1 << 20,), float(rank + 1), device=dev)
  dist.all_reduce(x)
  torch.cuda.synchronize()
  want = float(sum(range(1, world + 1)))
  assert float(x[0]) == want and float(x[-1]) == want, (float(x[0]), want)
  seen = [None] * world
  dist.all_gather_object(seen, {'rank': rank, 'pid': os.getpid()})
  assert [s['rank'] for s in seen] == list(range(world))
  dist.barrier()
  t0 = time.perf_counter()
  for _ in range(20):
    dist.all_reduce(x)
  torch.cuda.synchronize()
  print('[rank %d] all_reduce of 4 MiB: %.1f us each' % (rank, (time.perf_counter() - t0) / 20 * 1e6), flush=True)
  # the hand-off message on the device
  shape = (1, 256, 128)
  payload = torch.randn(shape, device=dev)
  if world == 1:
    msg = sharding.pack_handoff(payload, 7, 0)
    back = sharding.unpack_handoff(msg, 7, 0, shape)
    assert torch.equal(back, payload)
    try:
      sharding.unpack_handoff(msg, 8, 0, shape)
      raise AssertionError('a wrong song index went through')
    except sharding.HandoffError:
      pass
    print('[rank 0] hand-off message packed / header-checked in device memory: ok', flush=True)
  else:
    box = sharding._Outbox()
    if rank == 0:
      box.post(sharding.pack_handoff(payload, 3, 0), 1)
      box.drain()
      print('[rank 0] isend of the hand-off message completed', flush=True)
    else:
      buf = torch.empty(sharding.HEADER + 256 * 128, dtype=torch.float32, device=dev)
      sharding._recv(buf, 0)
      got = sharding.unpack_handoff(buf, 3, 0, shape)
      print('[rank 1] recv + header check of the hand-off message: ok (%d values)' % got.numel(), flush=True)
  dist.barrier()
  dist.destroy_process_group()
  print('[rank %d] done' % rank, flush=True)


def leg(world, deadline):
  port = free_port()
  procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), '--rank', str(r), str(world), str(port)],
                            stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(world)]
  t_end = time.time() + deadline
  outs, codes = [], []
  for p in procs:
    try:
      out, _ = p.communicate(timeout=max(1.0, t_end - time.time()))
    except subprocess.TimeoutExpired:
      p.kill()   # this exact child
      out, _ = p.communicate()
      out += '\n[probe] killed after the %d s deadline' % deadline
    outs.append(out)
    codes.append(p.returncode)
  return outs, codes


def tail(text, n=14):
  keep = [l for l in text.splitlines() if 'amdgpu.ids' not in l]
  return '\n'.join(keep[-n:])


def main():
  if len(sys.argv) > 1 and sys.argv[1] == '--rank':
    rank_main(int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]))
    return
  print('== leg 1: world 1 on RCCL (cuda:0)')
  outs, codes = leg(1, 240)
  print(tail(outs[0], 20))
  print('-> exit codes', codes)
  print('== leg 2: world 2, both ranks on cuda:0 (RCCL is expected to refuse)')
  outs, codes = leg(2, 180)
  for r, o in enumerate(outs):
    print('--- rank %d' % r)
    print(tail(o))
  print('-> exit codes', codes, '(0, 0 = the point-to-point hand-off ran over RCCL on one device)')


if __name__ == '__main__':
  main()
