"""Prompt templates used by ``chat()`` (host string code only).

Restates the one template the scorer uses — ``internlm2-chat`` — with the reference's semantics
(internvl/conversation.py:238-247 MPT join: ``system + sep`` then ``role + message + sep`` with NO
newline after the separator, an empty message leaves the bare role as the generation prompt;
template fields :371-387).  Other reference templates are out of scope.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import List, Optional, Tuple


@dataclass
class Conversation:
    name: str
    system_template: str = "{system_message}"
    system_message: str = ""
    roles: Tuple[str, str] = ("USER", "ASSISTANT")
    messages: List[List[Optional[str]]] = field(default_factory=list)
    sep: str = "\n"
    stop_token_ids: Optional[List[int]] = None

    def get_prompt(self) -> str:
        out = self.system_template.format(system_message=self.system_message) + self.sep
        for role, message in self.messages:
            if message:
                if isinstance(message, tuple):
                    message = message[0]
                out += role + message + self.sep
            else:
                out += role
        return out

    def set_system_message(self, system_message: str):
        self.system_message = system_message

    def append_message(self, role: str, message: Optional[str]):
        self.messages.append([role, message])

    def copy(self) -> "Conversation":
        return Conversation(self.name, self.system_template, self.system_message, self.roles,
                            [[r, m] for r, m in self.messages], self.sep, self.stop_token_ids)


_TEMPLATES = {
    "internlm2-chat": Conversation(
        name="internlm2-chat",
        system_template="<|im_start|>system\n{system_message}",
        system_message="你是由上海人工智能实验室联合商汤科技开发的书生多模态大模型，英文名叫InternVL, 是一个有用无害的人工智能助手。",
        roles=("<|im_start|>user\n", "<|im_start|>assistant\n"),
        sep="<|im_end|>",
        stop_token_ids=[2, 92543, 92542],
    ),
}


def get_conv_template(name: str) -> Conversation:
    if name not in _TEMPLATES:
        raise KeyError(f"conversation template '{name}' is not part of the scorer hot path (have: {list(_TEMPLATES)})")
    return _TEMPLATES[name].copy()
