"""Prompt templates with the reference's call surface (`conv_templates["llama3"].copy()`, `append_message`, `get_prompt`,
`roles`, `sep`; reference metamorph/conversation.py:26-140, 270-290; used by inference/demo.py:104-107 and train.py:508).

Rendered separator styles: LLAMA_3 (every shipped training / inference script), MPT (the chatml_direct / mistral_direct
templates) and PLAIN; asking for any other style raises.  The gradio / base64 helpers of the reference's class are
UI code and are not part of this package.  A message may be a `(text, image, mode)` tuple as in the reference: the text part is
what is rendered.  Prompts are pinned to the reference's by tests/golden/n2_conversation.json.
"""
from __future__ import annotations

import dataclasses
from enum import Enum, auto
from typing import List, Optional, Sequence

from .data import LLAMA3_ROLES, LLAMA3_SEP, LLAMA3_SYSTEM


class SeparatorStyle(Enum):
    SINGLE = auto()
    TWO = auto()
    MPT = auto()
    PLAIN = auto()
    LLAMA_2 = auto()
    LLAMA_3 = auto()


def _text(message):
    return message[0] if isinstance(message, tuple) else message


@dataclasses.dataclass
class Conversation:
    system: str
    roles: Sequence[str]
    messages: List[List[Optional[str]]]
    offset: int = 0
    sep_style: SeparatorStyle = SeparatorStyle.LLAMA_3
    sep: str = LLAMA3_SEP
    sep2: Optional[str] = None
    version: str = "Unknown"
    skip_next: bool = False

    def append_message(self, role, message):
        self.messages.append([role, message])

    def _messages_for_prompt(self):
        """A first message given as (text, image, mode) has its `<image>` marker moved to the front, on its own line."""
        msgs = [list(m) for m in self.messages]
        if msgs and isinstance(msgs[0][1], tuple):
            if "mmtag" in self.version:
                raise NotImplementedError("mmtag prompt versions are not used by MetaMorph")
            first = msgs[0][1][0].replace("<image>", "").strip()
            msgs[0][1] = "<image>\n" + first
        return msgs

    def get_prompt(self) -> str:
        if self.sep_style is SeparatorStyle.LLAMA_3:
            out = self.system                                # no separator after the system text
        elif self.sep_style in (SeparatorStyle.PLAIN, SeparatorStyle.MPT):
            out = self.system + self.sep
        else:
            raise ValueError(f"separator style {self.sep_style} is not rendered by metamorph_amd (llama3 / mpt / plain only)")
        for role, message in self._messages_for_prompt():
            out += role + _text(message) + self.sep if message else role
        return out

    def copy(self) -> "Conversation":
        return Conversation(system=self.system, roles=self.roles, messages=[[r, m] for r, m in self.messages], offset=self.offset,
                            sep_style=self.sep_style, sep=self.sep, sep2=self.sep2, version=self.version)

    def dict(self):
        return {"system": self.system, "roles": self.roles, "messages": [[r, _text(m)] for r, m in self.messages],
                "offset": self.offset, "sep": self.sep, "sep2": self.sep2}


conv_llama_3 = Conversation(system=LLAMA3_SYSTEM, roles=LLAMA3_ROLES, messages=[], offset=0, sep_style=SeparatorStyle.LLAMA_3,
                            sep=LLAMA3_SEP, version="llama3")
conv_chatml_direct = Conversation(system="", roles=("<|im_start|>user\n", "<|im_start|>assistant\n"), messages=[], offset=0,
                                  sep_style=SeparatorStyle.MPT, sep="<|im_end|>", version="mpt")

default_conversation = conv_llama_3
# the reference's registry also maps "default" / "v0" to a canned vicuna demo dialogue (SINGLE style); not carried here
conv_templates = {"llama3": conv_llama_3, "chatml_direct": conv_chatml_direct, "mistral_direct": conv_chatml_direct}
