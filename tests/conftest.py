import os
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in (HERE, ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(HERE, "golden")
GOLDEN_PAIRS = ["alice", "coding", "terror2", "plrabn12", "world192"]

# sha256 of the reference's committed vectors (reference README.md:6-19; test/*)
GOLDEN_SHA256 = {
    "alice.snappy": "30c025ef551ef55ba6120cdaf01fe9e321d4003a70b0d64bf30a56b0ddfe0f82",
    "alice.txt": "8b7e93cb820d71b2dd6387b2df5e7f488fc5b954c434092f8e24a22735c00f5a",
    "coding.snappy": "193407d63d31611b37bdb377483fd8b99690bfa9c6e7bb8a07976c162c399ef5",
    "coding.txt": "8b820d61a1c7a1e23f3cba75d144a5daf49aebc3d9e17b5b037d0b40389c28e5",
    "plrabn12.snappy": "8bc3e99e87d27fe2d3e47f8cc6c8b628d526a33514a45932e9fd5d2ec65722da",
    "plrabn12.txt": "07e2e0b461af78c7c647cb53dab39de560198e16f799b4516eccf0fbd69f764c",
    "terror2.snappy": "b19428398d1b6ed9eb0f35f07dcb69b2fa6b992ed4aa5555d27b317b578b707f",
    "terror2.txt": "49aaa4339923691a4f49e2ec3d5e5e250197bb418ab5f33aeb8d852d64f85d62",
    "world192.snappy": "edcb0875f83b2ea768ac6f7121191ab88caef3c8e14e477fa44d3119860c9625",
    "world192.txt": "c4c7862cdf18e8a39cb814f286dcd5f9aaf203bf6ff0bfd926ecc85d7ab24205",
    "xml.snappy": "eafe7445836d4746bcf9fc88c7008a738fd0f68a1c68cf8190e9379b43cfeda5",
}
# xml.txt is absent from the reference checkout (.MISSING_LARGE_BLOBS); SURVEY.md G5 records the
# digest of the plaintext its xml.snappy decodes to (5,345,280 bytes).
XML_TXT_SHA256 = "0e82e54e695c1938e4193448022543845b33020c8be6bf3bf3ead2224903e08c"
XML_TXT_LEN = 5345280


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def golden_bytes(name):
    with open(os.path.join(GOLDEN, name), "rb") as f:
        return f.read()


@pytest.fixture(scope="session")
def golden():
    return golden_bytes
