"""Piece-placement FEN from per-square labels (replaces python-chess' BaseBoard.set_piece_at / board_fen that
the reference uses at ``chessvision/core.py:330-336,342-349``; python-chess is not installed here)."""
from __future__ import annotations

from typing import Iterable

_PIECES = set("PNBRQKpnbrqk")


def board_fen(labels: Iterable[str], square_names: Iterable[str]) -> str:
    """labels[i] is the symbol on square_names[i] ("f" = empty).  Returns ranks 8..1 joined by "/"."""
    grid = [[None] * 8 for _ in range(8)]                 # grid[rank 0..7 = "1".."8"][file 0..7 = a..h]
    for label, name in zip(labels, square_names):
        file_i, rank_i = ord(name[0]) - ord("a"), int(name[1]) - 1
        if not (0 <= file_i < 8 and 0 <= rank_i < 8):
            raise ValueError(f"bad square name {name!r}")
        if label == "f":
            grid[rank_i][file_i] = None
        elif label in _PIECES:
            grid[rank_i][file_i] = label
        else:
            raise ValueError(f"invalid piece symbol: {label!r}")
    rows = []
    for rank_i in range(7, -1, -1):
        row, empty = "", 0
        for cell in grid[rank_i]:
            if cell is None:
                empty += 1
            else:
                row += (str(empty) if empty else "") + cell
                empty = 0
        rows.append(row + (str(empty) if empty else ""))
    return "/".join(rows)
