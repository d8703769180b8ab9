"""Keep N host cores busy for S seconds (a stand-in for a noisy neighbour on the GPU box's host): python scripts/cpu_hog.py N S"""
import multiprocessing as mp, sys, time
def spin(t_end):
    x = 0
    while time.time() < t_end:
        x += 1
if __name__ == "__main__":
    n, s = int(sys.argv[1]), float(sys.argv[2])
    t_end = time.time() + s
    ps = [mp.Process(target=spin, args=(t_end,)) for _ in range(n)]
    for p in ps: p.start()
    for p in ps: p.join()
