"""Builds the C oracle (oracle/oracle.c -> oracle/_build/liboracle_c.so) with gcc.

TEST INFRASTRUCTURE ONLY.  Called by __graft_entry__.build() and lazily by
oracle/c_oracle.py; the .so is git-ignored but travels to the GPU box with gpurun."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, 'oracle.c')
OUT_DIR = os.path.join(HERE, '_build')
OUT = os.path.join(OUT_DIR, 'liboracle_c.so')

FLAGS = ['-O2', '-fPIC', '-shared', '-std=gnu11', '-ffp-contract=off', '-fno-fast-math',
         '-fopenmp', '-Wall', '-Wno-unused-function']


def build(force=False):
    os.makedirs(OUT_DIR, exist_ok=True)
    if (not force and os.path.exists(OUT)
            and os.path.getmtime(OUT) >= os.path.getmtime(SRC)):
        return OUT
    tmp = OUT + '.tmp.%d' % os.getpid()
    subprocess.check_call(['gcc'] + FLAGS + [SRC, '-o', tmp, '-lm'])
    os.replace(tmp, OUT)
    return OUT


if __name__ == '__main__':
    print(build(force=True))
