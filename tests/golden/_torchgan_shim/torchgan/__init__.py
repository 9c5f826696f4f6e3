"""Fixture-generation shim ONLY (never shipped in the product path, never imported by tests).

torchgan==0.1.0 (requirements.txt:155 of the reference) is not installed in the build
container and cannot be fetched.  /root/reference/src/{wgan_loss,dcgan}.py subclass four of
its base classes; this package provides just those constructors (attributes only, no
arithmetic) so that the reference's own loss / train_ops code can be imported and run to
produce golden vectors.  See tests/golden/make_fixtures.py.
"""
