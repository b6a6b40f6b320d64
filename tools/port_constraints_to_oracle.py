"""Derives the oracle-side text of a block of constraint structs from its device-side text (developer helper).

The widened types are transcribed from the C# once, in the device form (velocity gate + pinned setup); the oracle form is the same
arithmetic without the gate. This script performs exactly that mechanical rewrite, so both sides come from one reading of the reference —
which is also why the oracle/device parity tests pin the GPU arithmetic, not the transcription (the behavioural tests in tests/test_oracle.py do that)."""
import re
import sys


def port(text: str) -> str:
    out = []
    for line in text.split("\n"):
        if re.match(r"\s*static constexpr int (wsA|access) = ", line):
            continue  # access filters are a device-side notion
        if re.match(r"\s*BD_GATE(_N)?\(", line) or re.match(r"\s*gate\(vA, vB\);", line):
            continue
        line = line.replace("template <class G> BD_FN", "static").replace(", G&& gate)", ")")
        line = line.replace("BD_FN", "static inline")
        out.append(line)
    return "\n".join(out)


if __name__ == "__main__":
    sys.stdout.write(port(open(sys.argv[1]).read()))
