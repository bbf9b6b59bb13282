"""Build oracle/_ref/: the reference's header-inline primitives, compiled from
/root/reference where the sources lie (nothing copied), through the harness
oracle/ref_primitives.pyx.  Only runs where /root/reference exists (this
container).  Outputs go to oracle/_ref/ only, which is git-ignored AND gpurun-ignored: nothing
compiled from the reference travels to the GPU box -- the pin that travels is the fixture this
build wrote, tests/golden/primitives.json."""
import os
import subprocess
import sys
import sysconfig

HERE = os.path.dirname(os.path.abspath(__file__))
REF = '/root/reference'
OUT = os.path.join(HERE, '_ref')


def build(force=False):
    if not os.path.isdir(os.path.join(REF, 'seekmer')):
        return None
    os.makedirs(OUT, exist_ok=True)
    ext = sysconfig.get_config_var('EXT_SUFFIX')
    so = os.path.join(OUT, 'ref_primitives' + ext)
    src = os.path.join(HERE, 'ref_primitives.pyx')
    if not force and os.path.exists(so) and os.path.getmtime(so) >= os.path.getmtime(src):
        return so
    c_file = os.path.join(OUT, 'ref_primitives.c')
    # legacy_implicit_noexcept: the semantics of the Cython 0.28 the reference pins
    # (environment.yml:9); results do not depend on it.
    subprocess.check_call([sys.executable, '-m', 'cython', '-3', '-I', REF,
                           '-X', 'legacy_implicit_noexcept=True',
                           src, '-o', c_file], stdout=subprocess.DEVNULL,
                          stderr=subprocess.DEVNULL)
    inc = sysconfig.get_paths()['include']
    subprocess.check_call(['gcc', '-O2', '-fPIC', '-shared', '-I', inc, c_file, '-o', so,
                           '-w'])
    return so


if __name__ == '__main__':
    print(build(force='--force' in sys.argv))
