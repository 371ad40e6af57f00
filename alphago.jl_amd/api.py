"""Host-side mirror of the reference's call surface for the hot path, over the C ABI.

Same names, argument meaning and error behaviour as tejank10/AlphaGo.jl so that the parity tests
read like the reference's own tests -- GoEnv, Position, play_move / pass_move / all_legal_moves /
score / result, NeuralNet, MCTSPlayer (initialize_game, tree_search, pick_move, play_move,
should_resign, is_done, set_result, extract_data), selfplay -- with Python conventions: 0-based
(row, col) coordinates, `None` for a pass, snake_case instead of `!`.  No arithmetic happens here:
every method marshals arrays and makes one call into libagz (HIP, gfx950)."""
from collections import namedtuple

import numpy as np

from . import _lib
from ._lib import IllegalMove
from .engine import Engine

BLACK, WHITE, EMPTY = 1, -1, 0
_KGS = "ABCDEFGHJKLMNOPQRST"
_SGF = "abcdefghijklmnopqrstuvwxyz"

PlayerMove = namedtuple("PlayerMove", "color move")       # board.jl:17-20

_rules = {}


def _rules_engine(N):
    """a 1-slot engine used only for the batched rule kernels (agz_go_*) with B = 1"""
    if N not in _rules:
        _rules[N] = Engine(board_size=N, tower_height=0, games=1, num_readouts=1, max_nodes_per_game=8)
    return _rules[N]


class GoEnv:
    """GoEnv(N, planes=17), src/game/go/go.jl:1-26"""

    def __init__(self, board_size=19, planes=17):
        assert planes % 2 == 1
        self.N = board_size
        self.action_space = board_size * board_size + 1
        self.planes = (planes - 1) // 2
        self.max_action_space = 361          # go.jl:24 (hard-coded in the reference)


def to_flat(c, env):                          # coords.jl:5-7
    return env.N * env.N if c is None else c[0] + env.N * c[1]


def from_flat(f, env):                        # coords.jl:10-12
    return None if f == env.N * env.N else (f % env.N, f // env.N)


def from_kgs(s, env):                         # coords.jl:26-34
    if s == "pass":
        return None
    return env.N - int(s[1:]), _KGS.index(s[0].upper())


def to_kgs(c, env):                           # coords.jl:37
    return "pass" if c is None else f"{_KGS[c[1]]}{env.N - c[0]}"


def from_sgf(s):                              # coords.jl:14-20
    return None if not s else (_SGF.index(s[1]), _SGF.index(s[0]))


def to_sgf(c):                                # coords.jl:23
    return "" if c is None else _SGF[c[1]] + _SGF[c[0]]


class Position:
    """GoPosition, src/game/go/board.jl:271-306.  board[row, col] in {-1, 0, +1}."""

    def __init__(self, env, board=None, n=0, komi=7.5, caps=(0, 0), ko=None, recent=(), board_deltas=None,
                 to_play=BLACK):
        self.env = env
        N = env.N
        self.board = np.zeros((N, N), np.int8) if board is None else np.array(board, np.int8).reshape(N, N)
        self.n = n
        self.komi = float(np.float32(komi))
        self.caps = tuple(caps)
        self.ko = ko
        self.recent = list(recent)
        self.board_deltas = np.zeros((0, N, N), np.int8) if board_deltas is None else np.array(board_deltas, np.int8)
        self.to_play = to_play
        self.done = False

    # flat views in the ABI's point order p = row + N*col
    def _flat(self):
        return np.ascontiguousarray(self.board.T).reshape(1, -1)

    def _ko0(self):
        return -1 if self.ko is None else to_flat(self.ko, self.env)

    def soa(self):
        """(board [P], deltas [7, P], ndeltas, to_play) for agz_net_forward / agz_features"""
        N = self.env.N
        d = np.zeros((7, N * N), np.int8)
        k = self.board_deltas.shape[0]
        for i in range(k):
            d[i] = np.ascontiguousarray(self.board_deltas[i].T).reshape(-1)
        return self._flat()[0], d, k, self.to_play

    def all_legal_moves(self):                # board.jl:393-424
        return _rules_engine(self.env.N).go_legal(self._flat(), [self.to_play], [self._ko0()])[0]

    def is_move_legal(self, c):               # board.jl:376-391
        return bool(self.all_legal_moves()[to_flat(c, self.env)])

    def score(self):                          # board.jl:511-533
        return float(_rules_engine(self.env.N).go_score(self._flat(), [self.komi])[0])

    def result(self):                         # board.jl:535-544
        s = self.score()
        return 1 if s > 0 else -1 if s < 0 else 0

    def result_string(self):                  # board.jl:546-555
        s = self.score()
        return f"B+{s:.1f}" if s > 0 else f"W+{-s:.1f}" if s < 0 else "DRAW"

    def _copy(self):
        p = Position(self.env, self.board.copy(), self.n, self.komi, self.caps, self.ko, list(self.recent),
                     self.board_deltas.copy(), self.to_play)
        return p                              # like deepcopy(GoPosition): done resets (board.jl:308-315)

    def pass_move(self, mutate=False):        # board.jl:426-440
        return self.play_move(None, mutate=mutate)

    def flip_playerturn(self, mutate=False):  # board.jl:442-447
        p = self if mutate else self._copy()
        p.ko = None
        p.to_play = -p.to_play
        return p

    def play_move(self, c, mutate=False):     # board.jl:451-509
        env, N = self.env, self.env.N
        a = to_flat(c, env)
        bo, ko_o, nc, st = _rules_engine(N).go_play(self._flat(), [self.to_play], [self._ko0()], [a])
        if st[0] == _lib.ILLEGAL_MOVE:
            raise IllegalMove(_lib.ILLEGAL_MOVE, f"illegal move {c}")
        new_board = bo[0].reshape(N, N).T.copy()
        color = self.to_play
        # delta = +color at the played point and wherever an opponent stone vanished (board.jl:479-481)
        delta = np.where(new_board != self.board, color, 0).astype(np.int8)
        p = self if mutate else self._copy()
        was_pass_before = bool(self.recent) and self.recent[-1].move is None
        p.board = new_board
        p.n = self.n + 1
        cap = int(nc[0])
        p.caps = (self.caps[0] + cap, self.caps[1]) if color == BLACK else (self.caps[0], self.caps[1] + cap)
        p.ko = None if ko_o[0] < 0 else from_flat(int(ko_o[0]), env)
        p.recent = list(self.recent) + [PlayerMove(color, c)]
        keep = self.board_deltas[: env.planes - 2]
        p.board_deltas = np.concatenate([delta[None], keep], axis=0)
        p.to_play = -color
        p.done = c is None and was_pass_before
        return p


class NeuralNet:
    """NeuralNet(env; tower_height), src/neural_net.jl:13-33 -- inference only"""

    def __init__(self, env, tower_height=19, seed=0):
        self.env = env
        self.tower_height = tower_height
        self.engine = Engine(board_size=env.N, tower_height=tower_height, games=1, num_readouts=1,
                             max_nodes_per_game=8)
        self.engine.init_synthetic(seed)      # Flux-default-equivalent init (glorot uniform, BN identity)

    def set_weights(self, layer, kind, data):
        self.engine.set_weights(layer, kind, data)

    def set_precision(self, precision="f32"):
        """tower arithmetic of nn(positions): "f32" (default) or "f16" (fp16 operands, f32 accumulate)"""
        self.engine.set_precision(precision)

    def __call__(self, positions):            # neural_net.jl:57-73
        single = isinstance(positions, Position)
        plist = [positions] if single else list(positions)
        if plist and not isinstance(plist[0], Position):
            feats = np.stack([np.asarray(p.feats, np.float32).reshape(-1) for p in plist])
            pi, v = self.engine.forward_features(feats)
        else:
            soa = [p.soa() for p in plist]
            pi, v = self.engine.forward(np.stack([s[0] for s in soa]), np.stack([s[1] for s in soa]),
                                        [s[2] for s in soa], [s[3] for s in soa])
        return (pi[0], float(v[0])) if single else (pi.T.copy(), v)      # pi is A x B like the reference

    def forward_features(self, feats):
        return self.engine.forward_features(feats)


def load_model(model_dir, env, bn_field="auto"):
    """load_model(str, env), src/play.jl:3-21: BSON parameter lists (and BatchNorm statistics when the
    struct dumps are present) -> NeuralNet; the tower height is read off the base parameter list.
    bn_field says what the 5th field of a dumped Flux.BatchNorm is: "std" (Flux <= 0.7, the files the
    reference ships: forward (x - mu) / sigma), "var" (Flux >= 0.8: sigma^2, eps under the root) or
    "auto" (decide per layer from the dump: Float64 eps = 1e-8 / TrackedArray parameters => "std")."""
    from . import bson_weights as bw
    ck = bw.read_checkpoint(model_dir, bn_field)
    nn = NeuralNet(env, tower_height=bw.tower_height_of(ck["base"]))
    bw.apply_param_lists(nn.engine, ck["base"], ck["value"], ck["policy"], ck.get("base_stats"),
                         ck.get("value_stats"), ck.get("policy_stats"))
    return nn


def save_model(nn, model_dir):
    """save_model(nn), src/train.jl:14-35 (the reference hard-codes <repo>/models; here the directory
    is an argument): writes weights/agz_{base,value,policy}.bson readable by Flux.loadparams!"""
    from . import bson_weights as bw
    bw.write_checkpoint(model_dir, bw.extract_param_lists(nn.engine))


def get_feats(pos):                           # features.jl:24-26 -> [17, N, N] indexed [plane, row, col]
    N = pos.env.N
    b, d, k, tp = pos.soa()
    f = _rules_engine(N).features(b[None], d[None], [k], [tp])[0]
    return f.reshape(17, N, N).transpose(0, 2, 1).copy()


class LeafPosition(Position):
    """One element of the `Vector{Position}` a duck-typed network receives (`mcts_player.network([leaf.position for leaf
    in leaves])`, mcts_play.jl:89): the leaf's GoPosition fields materialised from the device tree by
    agz_tree_leaf_positions -- board, board_deltas (newest first), to_play, n, ko, caps, the last two moves -- so that
    `len(positions)` is the batch size (DummyNet, test/test_mcts_player.jl:25-32) and get_feats(position) /
    NeuralNet(positions) work on it like on any Position.  `.node` is the leaf's handle; `.feats` the 17 planes
    (plane-major, p = row + N*col), computed on demand by agz_features."""

    node = None

    @property
    def feats(self):
        b, d, k, tp = self.soa()
        return _rules_engine(self.env.N).features(b[None], d[None], [k], [tp])[0].reshape(-1)


LeafView = LeafPosition       # round <= 5 name


def _leaf_positions(player, lp):
    env, N = player.env, player.env.N
    out = []
    for k in range(len(lp["nodes"])):
        i = lp["info"][k]
        nd = int(lp["ndeltas"][k])
        recent = []
        if i.prev_move >= 0:
            recent.append(PlayerMove(int(i.to_play), from_flat(int(i.prev_move), env)))
        if i.last_move >= 0:
            recent.append(PlayerMove(-int(i.to_play), from_flat(int(i.last_move), env)))
        pos = LeafPosition(env, lp["boards"][k].reshape(N, N).T, int(i.n), i.komi, (int(i.caps_black), int(i.caps_white)),
                           None if i.ko < 0 else from_flat(int(i.ko), env), recent,
                           lp["deltas"][k, :nd].reshape(nd, N, N).transpose(0, 2, 1), int(i.to_play))
        pos.node = NodeView(player, int(lp["nodes"][k]))
        out.append(pos)
    return out


class NodeView:
    """read-only view of an MCTSNode (src/mcts.jl:41-82) living on the device"""

    def __init__(self, player, node):
        self._p, self.id = player, node

    @property
    def _info(self):
        return self._p.engine.node_info(0, self.id)

    N = property(lambda s: s._info.N)
    W = property(lambda s: s._info.W)
    Q = property(lambda s: s._info.Q)
    is_expanded = property(lambda s: bool(s._info.is_expanded))
    losses_applied = property(lambda s: s._info.losses_applied)
    fmove = property(lambda s: s._info.fmove)
    child_N = property(lambda s: s._p.engine.node_floats(0, s.id, _lib.F_CHILD_N))
    child_W = property(lambda s: s._p.engine.node_floats(0, s.id, _lib.F_CHILD_W))
    child_prior = property(lambda s: s._p.engine.node_floats(0, s.id, _lib.F_CHILD_PRIOR))
    child_action_score = property(lambda s: s._p.engine.node_scores(0, s.id))

    @property
    def child_Q(self):
        return self.child_W / (np.float32(1) + self.child_N)

    @property
    def child_U(self):                        # mcts.jl:91-92: c_puct (Float64) x Float32 sqrt x prior / (1 + N)
        # Julia evaluates left to right: Float64 c_puct x the Float32 square root, then Float64 throughout
        scale = np.float64(self._p.engine.cfg.c_puct) * np.float64(np.sqrt(np.float32(1) + np.float32(self.N)))
        return scale * self.child_prior.astype(np.float64) / (np.float32(1) + self.child_N).astype(np.float64)

    def __eq__(self, other):
        return isinstance(other, NodeView) and other._p is self._p and other.id == self.id

    def __hash__(self):
        return hash((id(self._p), self.id))

    # ---- the node-level operations test/test_mcts.jl:2-5 imports, one agz_tree_* call each
    def select_leaf(self):                    # mcts.jl:108-138
        return NodeView(self._p, self._p.engine.select_leaf(0, self.id))

    def maybe_add_child(self, f):             # mcts.jl:140-149 (f: 0-based flat move, N*N = pass)
        return NodeView(self._p, self._p.engine.maybe_add_child(0, self.id, int(f)))

    def add_virtual_loss(self, up_to):        # mcts.jl:151-163
        self._p.engine.add_virtual_loss(0, self.id, up_to.id)

    def revert_virtual_loss(self, up_to):     # mcts.jl:165-177
        self._p.engine.revert_virtual_loss(0, self.id, up_to.id)

    def incorporate_results(self, move_probs, value, up_to):      # mcts.jl:187-214
        e = self._p.engine
        st = e.incorporate_results(0, self.id, np.asarray(move_probs, np.float32), float(value), up_to.id)
        if st in (_lib.ASSERT_DONE_NODE, _lib.BAD_SHAPE):
            raise AssertionError(e.L.agz_last_error(e.h).decode())      # mcts.jl:190,196
        e._ck(st)

    def inject_noise(self):                   # mcts.jl:232-239
        self._p.engine.inject_noise(0, self.id)

    def set_N(self, value):                   # mcts.jl:99
        self._p.engine.node_set_N(0, self.id, float(value))

    def is_done(self):                        # mcts.jl:227-229
        return bool(self._p.engine.is_done(0, self.id))

    @property
    def children(self):
        ch = self._p.engine.node_children(0, self.id)
        return {int(a): NodeView(self._p, int(c)) for a, c in enumerate(ch) if c >= 0}

    @property
    def position(self):
        info = self._info
        N = self._p.env.N
        board = self._p.engine.node_board(0, self.id).reshape(N, N).T
        pos = Position(self._p.env, board, info.pos.n, info.pos.komi, (info.pos.caps_black, info.pos.caps_white),
                       None if info.pos.ko < 0 else from_flat(info.pos.ko, self._p.env), to_play=info.pos.to_play)
        pos.done = bool(info.done)
        # `recent`: the whole move list when this node is the player's root (what extract_data / replay_position need,
        # board.jl:557-578); for any other node the moves played since the root, behind the root's
        moves, node, inf = [], self.id, info
        while inf.parent >= 0:
            moves.append(inf.fmove)
            node = inf.parent
            inf = self._p.engine.node_info(0, node)
        base = getattr(self._p, "_recent", None)
        if base is not None and node == self._p.engine.tree_root(0):
            rec, tp = list(base), -info.pos.to_play if len(moves) % 2 else info.pos.to_play
            for a in reversed(moves):
                rec.append(PlayerMove(tp, from_flat(int(a), self._p.env)))
                tp = -tp
            pos.recent = rec
        return pos


class MCTSPlayer:
    """MCTSPlayer(env, network; num_readouts, two_player_mode, resign_threshold), mcts_play.jl:3-24.
    `network` is any callable positions -> (pi A x B, v B) (the duck-typed field of mcts_play.jl:5): a NeuralNet of this
    package is evaluated on the device; anything else receives a list of B Position objects (LeafPosition) exactly as
    the reference's `mcts_player.network([leaf.position for leaf in leaves])` (mcts_play.jl:89) and may answer with
    plain arrays or Tracker-style objects carrying `.data` (mcts_play.jl:90)."""

    def __init__(self, env, network, num_readouts=800, two_player_mode=False, resign_threshold=-0.9, seed=0,
                 game_id=0):
        self.env = env
        self.network = network
        self.num_readouts = num_readouts
        self.two_player_mode = two_player_mode
        self.tau_threshold = -1 if two_player_mode else (env.N * env.N // 12) // 2 * 2
        self.resign_threshold = resign_threshold
        internal = isinstance(network, NeuralNet)
        self.engine = Engine(board_size=env.N, tower_height=network.tower_height if internal else 0, games=1,
                             num_readouts=num_readouts, parallel_readouts=64, two_player_mode=int(two_player_mode),
                             resign_threshold=resign_threshold, seed=seed, external_network=0 if internal else 1)
        if internal:
            network.engine.copy_weights_to(self.engine)
        self._game_id = game_id
        self.qs, self.searches_pi = [], []
        self.result, self.result_string = 0, ""
        self._start = None

    def initialize_game(self, pos=None):      # mcts_play.jl:110-118
        pos = Position(self.env) if pos is None else pos
        last = -1 if not pos.recent else to_flat(pos.recent[-1].move, self.env)
        # initialize_game!(player, pos) keeps pos.board_deltas (board.jl:505-506): the up-to-7 older boards the
        # history planes need are B_{k+1} = B_k - delta_k (features.jl:8-14), newest first
        hist, b = [], pos._flat()[0].astype(np.int16)
        for k in range(min(7, pos.board_deltas.shape[0])):
            b = b - np.ascontiguousarray(pos.board_deltas[k].T).reshape(-1)
            hist.append(b.astype(np.int8))
        self.engine.tree_init(0, pos._flat()[0], n=pos.n, to_play=pos.to_play, ko=pos._ko0(), caps=pos.caps,
                              last_move=last, komi=pos.komi, history=np.stack(hist) if hist else None)
        self.engine.set_draw(0, self._game_id, 0)
        self.qs, self.searches_pi = [], []
        self.result, self.result_string = 0, ""
        self._start = pos
        self._moves = []
        self._recent = list(pos.recent)

    @property
    def root(self):
        return NodeView(self, self.engine.tree_root(0))

    def tree_search(self, parallel_readouts=8):    # mcts_play.jl:73-98; returns the leaves like the reference
        e = self.engine
        n = e.tree_search_select(0, parallel_readouts)
        if isinstance(self.network, NeuralNet):
            nodes = e.tree_leaf_positions(0, n, nodes_only=True)["nodes"]
            e.tree_search_incorporate(0)
            return [NodeView(self, int(i)) for i in nodes]
        if n == 0:
            e.tree_search_incorporate(0)
            return []
        positions = _leaf_positions(self, e.tree_leaf_positions(0, n))
        move_probs, values = self.network(positions)             # mcts_play.jl:89
        move_probs = np.asarray(getattr(move_probs, "data", move_probs), np.float32)      # :90
        values = np.asarray(getattr(values, "data", values), np.float32).reshape(-1)
        if move_probs.shape != (self.env.action_space, n) or values.shape != (n,):
            raise AssertionError(f"network returned {move_probs.shape} / {values.shape} for {n} positions "
                                 f"(expected ({self.env.action_space}, {n}) / ({n},))")      # mcts.jl:190
        e.tree_search_incorporate(0, np.ascontiguousarray(move_probs.T), values)          # column i = leaf i (:91)
        return [p.node for p in positions]

    def pick_move(self):                      # mcts_play.jl:52-71
        st, a = self.engine.pick_move(0)
        if st == _lib.ASSERT_SOFTPICK:
            raise AssertionError("child_N[fcoord] != 0 (mcts_play.jl:67)")
        return from_flat(a, self.env)

    def play_move(self, c):                   # mcts_play.jl:26-50
        root = self.root
        info = root._info
        if not self.two_player_mode:
            cn = root.child_N
            with np.errstate(invalid="ignore", divide="ignore"):
                if info.pos.n <= self.tau_threshold:
                    pr = cn.astype(np.float64) ** 0.98
                    self.searches_pi.append((pr / pr.sum()).astype(np.float32))
                else:
                    self.searches_pi.append(cn / cn.sum())
        self.qs.append(np.float32(info.Q))
        if not self.engine.play_move(0, to_flat(c, self.env)):
            print("Illegal move")
            if not self.two_player_mode:
                self.searches_pi.pop()
            self.qs.pop()
            return False
        self._moves.append(c)
        self._recent.append(PlayerMove(info.pos.to_play, c))
        return True

    def get_position(self):                   # mcts_play.jl:141-142
        return self.root.position

    def suggest_move(self):                   # mcts_play.jl:144-151
        current_readouts = self.root.N
        while self.root.N < current_readouts + self.num_readouts:
            self.tree_search()
        return self.pick_move()

    def should_resign(self):                  # mcts_play.jl:124
        return bool(self.engine.should_resign(0))

    def is_done(self):                        # mcts_play.jl:120
        return self.result != 0 or bool(self.engine.is_done(0, self.engine.tree_root(0)))

    def set_result(self, winner, was_resign):  # mcts_play.jl:100-108
        self.result = winner
        self.result_string = ("B+R" if winner == BLACK else "W+R") if was_resign else self.root.position.result_string()

    def extract_data(self):                   # mcts_play.jl:126-139
        assert len(self.searches_pi) == self.root._info.pos.n, "GoPosition history is incomplete"
        pos = Position(self.env, komi=self._start.komi)
        positions = []
        for c in self._moves:
            positions.append(pos)
            pos = pos.play_move(c)
        return positions, [p.copy() for p in self.searches_pi], [self.result] * len(positions)


# A bare finished-game tuple (what records look like before they are wrapped; tests and the replay buffer build them by
# hand).  short_searches: moves played on fewer than num_ro readouts because the node pool was full (0 = the game is the
# reference's game; agz_config.pool_policy, include/agz.h)
GameRecord = namedtuple("GameRecord", "game_id moves searches_pi qs result result_string was_resign short_searches",
                        defaults=[0])


class _FinishedRoot:
    """`player.root` of a finished self-play game: the one thing train() reads from it is `.position`
    (train.jl:72: player.root.position.n; mcts_play.jl:127,132: extract_data replays root.position.recent)"""

    def __init__(self, player):
        self._player = player

    @property
    def position(self):
        return self._player._replay()[1]


class SelfPlayPlayer:
    """What `selfplay(env, nn, num_ro)` returns (selfplay.jl:44): the MCTSPlayer of ONE finished game, read-only --
    `.result`, `.result_string`, `.qs`, `.searches_pi` (mcts_play.jl:3-15) as the device recorded them,
    `.root.position` (the final GoPosition with its whole `recent` list: `.n`, `.board`, `.caps`, ...) and
    `extract_data(player)`.  The tree itself stayed on the device and was recycled with its slot.  Also carries the
    record's fields (`game_id`, `moves` as board coordinates / None, `was_resign`, `short_searches`)."""

    def __init__(self, env, network, num_readouts, rec):
        self.env, self.network, self.num_readouts = env, network, num_readouts
        self.two_player_mode = False
        self.tau_threshold = (env.N * env.N // 12) // 2 * 2
        self.game_id = int(rec["game_id"])
        self.resign_threshold = -1.0 if rec.get("resign_disabled") else -0.9       # selfplay.jl:9
        self.moves = [from_flat(int(a), env) for a in rec["moves"]]
        self.searches_pi = [np.array(p, np.float32) for p in rec["pis"]]
        self.qs = np.array(rec["qs"], np.float32)
        self.result = int(rec["result"])
        self.was_resign = bool(rec["was_resign"])
        self.short_searches = int(rec.get("short_searches", 0))
        if self.was_resign:                                                         # mcts_play.jl:100-108
            self.result_string = "B+R" if self.result == BLACK else "W+R"
        else:
            sc = float(rec["final_score"])                                          # board.jl:546-555
            self.result_string = f"B+{sc:.1f}" if sc > 0 else f"W+{-sc:.1f}" if sc < 0 else "DRAW"
        self.root = _FinishedRoot(self)
        self._replayed = None

    def _replay(self):
        """replay_position (board.jl:557-578): the positions before each move and the final one, by agz_go_play"""
        if self._replayed is None:
            pos, before = Position(self.env), []
            for c in self.moves:
                before.append(pos)
                pos = pos.play_move(c)
            self._replayed = (before, pos)
        return self._replayed

    @property
    def position(self):                        # mcts_play.jl:14
        return self.root.position

    def get_position(self):                    # mcts_play.jl:141-142
        return self.root.position

    def is_done(self):                         # mcts_play.jl:120
        return True

    def extract_data(self):                    # mcts_play.jl:126-139
        assert len(self.searches_pi) == self.root.position.n, "GoPosition history is incomplete"
        before, _ = self._replay()
        return list(before), [p.copy() for p in self.searches_pi], [self.result] * len(before)


# The reference draws from Julia's global RNG (selfplay.jl:9, mcts.jl:133,235, mcts_play.jl:61,66): successive selfplay
# calls see successive random numbers.  Here every draw is a function of (seed, game id, move, site) (include/agz_draws.h):
# `seed(s)` is Random.seed!(s), and each selfplay call plays the next unused game ids of that stream.
_stream = {"seed": 0, "next_game": 0}


def seed(s):
    """Random.seed!(s) for selfplay(): restarts the game-id stream at 0 under draw-stream seed `s`"""
    _stream["seed"], _stream["next_game"] = int(s), 0


def selfplay(env, nn, num_ro=800, games=None, seed=None, slots=None, precision="f32", game_id_base=None, **cfg):
    """selfplay(env, nn, num_ro) (src/selfplay.jl:1-45) -> the finished game's player (SelfPlayPlayer), exactly the
    call train() makes (train.jl:57).  `games=G` (ours) plays G games concurrently on the device and returns a list of
    G such players ordered by game id.  Game ids continue from the previous call (module stream, `seed()`), unless
    `seed` / `game_id_base` pin them.  precision="f16" plays with the fp16-operand tower; default exact f32."""
    single = games is None
    games = 1 if single else int(games)
    if seed is None:
        seed = _stream["seed"]
        if game_id_base is None:
            game_id_base = _stream["next_game"]
            _stream["next_game"] += games
    if game_id_base is None:
        game_id_base = 0
    slots = min(games, 1024) if slots is None else slots
    eng = Engine(board_size=env.N, tower_height=nn.tower_height, games=slots, num_readouts=num_ro, seed=seed,
                 game_id_base=game_id_base, record_capacity_games=games + 8, **cfg)
    nn.engine.copy_weights_to(eng)
    eng.set_precision(precision)
    eng.start(games)
    while eng.records_count() < games:
        eng.step(16)
        if eng.stats()["stalled_games"]:      # only with pool_policy = AGZ_POOL_STALL: a game waits on its full pool
            eng.close()
            raise _lib.AgzError(_lib.POOL_EXHAUSTED, "a game is waiting on a full node pool (pool_policy = stall): raise "
                                                     "max_nodes_per_game or use the default policy")
    out = [SelfPlayPlayer(env, nn, num_ro, r) for r in eng.records()]
    eng.close()
    return out[0] if single else out


EvalStats = namedtuple("EvalStats", "games_won num_games win_rate resigned moves records")


def evaluate(env, black_net, white_net, num_games=400, ro=800, verbose=False, seed=0, slots=None,
             return_stats=False, **cfg):
    """evaluate(env, black_net, white_net; num_games, ro) (src/neural_net.jl:103-158): black_net plays
    Black and white_net White in `num_games` games of two two_player_mode MCTSPlayers (arg-max moves,
    no noise, resign at -0.9); True iff Black's win rate reaches 0.55.  All games run concurrently on
    the device (arena_mode: one slot pair per game, both networks resident).  The tally follows the
    reference literally: a game counts for Black when `result(black.root.position) == BLACK`, i.e.
    by the Tromp-Taylor score of the final position, also after a resignation (:147)."""
    if black_net.tower_height != white_net.tower_height:
        raise ValueError("the arena keeps both networks in one engine: tower heights must match")
    pairs = min(num_games, 512) if slots is None else slots
    eng = Engine(board_size=env.N, tower_height=black_net.tower_height, games=2 * pairs, num_readouts=ro, seed=seed,
                 arena_mode=1, record_capacity_games=num_games + 8, **cfg)
    black_net.engine.copy_weights_to(eng)
    eng.net_select(1)
    white_net.engine.copy_weights_to(eng)
    eng.net_select(0)
    eng.start(num_games)
    while eng.records_count() < num_games:
        eng.step(16)
        if eng.stats()["pool_exhausted"]:
            eng.close()
            raise _lib.AgzError(_lib.POOL_EXHAUSTED, "node pool exhausted; raise max_nodes_per_game")
    recs = eng.records()
    st = eng.stats()
    eng.close()
    if st["pool_exhausted"]:
        raise _lib.AgzError(_lib.POOL_EXHAUSTED, "node pool exhausted; raise max_nodes_per_game")
    games_won = sum(1 for r in recs if r["final_score"] > 0)
    rate = games_won / num_games
    if verbose:
        print(f"Won {games_won} / {num_games}. Win rate: {rate}. ", end="")
    ok = rate >= 0.55
    if return_stats:
        return ok, EvalStats(games_won, num_games, rate, sum(int(r["was_resign"]) for r in recs),
                             sum(int(r["num_moves"]) for r in recs), recs)
    return ok


def extract_data(player, record=None):
    """extract_data(player) -> (positions, pis, results), mcts_play.jl:126-139: one argument, the player selfplay()
    returned or a live MCTSPlayer (train.jl:58).  The round <= 5 form extract_data(env, record) for a bare GameRecord
    is still accepted."""
    if record is None:
        return player.extract_data()
    env, pos, positions = player, Position(player), []
    for c in record.moves:
        positions.append(pos)
        pos = pos.play_move(c)
    return positions, [np.array(p) for p in record.searches_pi], [record.result] * len(positions)


def get_replay_batch(pos_buffer, pi_buffer, res_buffer, batch_size=32, rng=None):
    """get_replay_batch(pos_buffer, pi_buffer, res_buffer; batch_size), src/train.jl:4-12: `batch_size` distinct
    entries (sample(..., replace=false)); pi_replay = hcat(...) is A x B.  (alphago.jl_amd.ReplayBuffer is the same
    contract with positions kept as move lists and rebuilt on the device.)"""
    rng = np.random.default_rng() if rng is None else rng
    idxs = rng.choice(len(pos_buffer), size=batch_size, replace=False)
    return ([pos_buffer[i] for i in idxs], np.stack([pi_buffer[i] for i in idxs], axis=1),
            [res_buffer[i] for i in idxs])


class Momentum:
    """Flux.Momentum(eta, rho = 0.9) (train.jl:54): the state lives in the network's engine (agz_train_step)"""

    def __init__(self, eta=0.01, rho=0.9):
        self.eta, self.rho = float(eta), float(rho)


def _train(nn, input_data, opt, epochs=1):
    """_train(nn, (positions, pi A x B, z), opt; epochs) (src/neural_net.jl:85-101; call train.jl:70) as intended (the
    reference's does not run at HEAD): minibatches of 32 positions (a short tail is its own batch; a single
    left-over position joins the batch before it: BatchNorm needs two), each one agz_train_step on the device --
    training-mode forward, 0.01 crossentropy + 0.01 mse + 1e-4 sum(theta^2), backward, Momentum update.  Returns the
    summed minibatch loss / epochs (:98-100).  Features come from agz_features on the positions' own fields."""
    positions, pi, z = input_data
    pi = np.asarray(pi, np.float32)
    z = np.asarray(z, np.float32)
    e = nn.engine
    soa = [p.soa() for p in positions]
    feats = e.features(np.stack([s[0] for s in soa]), np.stack([s[1] for s in soa]), [s[2] for s in soa],
                       [s[3] for s in soa])
    n = len(positions)
    cuts = list(range(0, n, 32)) + [n]
    if len(cuts) > 2 and cuts[-1] - cuts[-2] == 1:
        del cuts[-2]
    loss_avg = 0.0
    for _ in range(epochs):
        for lo, hi in zip(cuts[:-1], cuts[1:]):
            loss_avg += float(e.train_step(feats[lo:hi], pi[:, lo:hi].T, z[lo:hi], eta=opt.eta, rho=opt.rho)[0])
    return loss_avg / epochs
