"""Import stub (absent offline); nothing on the pinned path calls into it."""


class AudioSegment:  # noqa: D101
    pass
