from .rpn import PROPOSAL_GENERATOR_REGISTRY, RPN, RPN_HEAD_REGISTRY, RPNWNM, RRPN, StandardRPNHead, build_proposal_generator
