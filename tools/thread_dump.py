"""Diagnostic aid for a stalled multi-rank rehearsal: after `after_s` seconds print, for every OS thread of this process,
its name, scheduler state, the kernel function it sleeps in (wchan) and its current system call -- enough to tell a
thread blocked in a GPU wait (ioctl / busy loop in libhsa) from one blocked in gloo's TCP transport (epoll/recv) or on
a condition variable (futex), without a debugger."""
import os
import sys
import threading
import time


def _read(path):
    try:
        with open(path) as f:
            return f.read().strip()
    except OSError as e:
        return f"<{e.strerror}>"


def dump(tag=""):
    pid = os.getpid()
    lines = [f"---- thread dump {tag} pid {pid} rank {os.environ.get('RANK', '?')} ----"]
    for tid in sorted(os.listdir(f"/proc/{pid}/task"), key=int):
        base = f"/proc/{pid}/task/{tid}"
        stat = _read(base + "/stat")
        state = stat.rsplit(")", 1)[-1].split()[0] if ")" in stat else "?"
        utime = stat.rsplit(")", 1)[-1].split()[11:13] if ")" in stat else []
        lines.append(f"tid {tid:>7} {_read(base + '/comm'):<18} state {state} cpu(u,s) {','.join(utime):<12} "
                     f"wchan {_read(base + '/wchan'):<28} syscall {_read(base + '/syscall')[:60]}")
    sys.stderr.write("\n".join(lines) + "\n")
    sys.stderr.flush()


def _native():
    """tools/libnative_bt.so (native_bt.c), if it has been built: native backtraces by signal"""
    import ctypes
    so = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libnative_bt.so")
    if not os.path.exists(so):
        return None
    lib = ctypes.CDLL(so)
    return lib if lib.nbt_install() == 0 else None


def native_stacks(lib, only_busy=True):
    """the native stack of every thread that has burnt CPU (or of all threads), one after the other"""
    pid = os.getpid()
    for tid in sorted(os.listdir(f"/proc/{pid}/task"), key=int):
        stat = _read(f"/proc/{pid}/task/{tid}/stat")
        f = stat.rsplit(")", 1)[-1].split() if ")" in stat else []
        busy = len(f) > 12 and int(f[11]) + int(f[12]) > 200  # > 2 s of CPU
        if busy or not only_busy:
            lib.nbt_signal(int(tid))
            time.sleep(0.2)


def arm(after_s, repeat=2, gap_s=2.0):
    lib = _native()

    def run():
        time.sleep(after_s)
        for i in range(repeat):
            dump(f"#{i}")
            if lib is not None:
                native_stacks(lib, only_busy=(i == 0))
            time.sleep(gap_s)
    t = threading.Thread(target=run, daemon=True, name="thread_dump")
    t.start()
    return t
