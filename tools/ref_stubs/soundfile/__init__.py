"""Import stub (absent offline).  `read` hands back an array the golden script planted (tools/gen_goldens_hubert.py calls
the reference's own process_audio with its file I/O mocked); no arithmetic."""
_data = None


def read(path):
    return _data, 16000
