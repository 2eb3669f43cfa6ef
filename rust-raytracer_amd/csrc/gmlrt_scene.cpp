#include "gmlrt_scene.hpp"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <map>
#include <tuple>

#include "gmlrt_device.hpp"
#include "gmlrt_surface_vm.h"

namespace gmlrt {

// ============================================================== symbolic values
namespace {

struct Expr;
using ExprP = std::shared_ptr<Expr>;
struct Expr {
  enum Ty : uint8_t { I, R, B } ty;
  bool is_const = false;
  uint64_t bits = 0;   // constant payload: i64, f64 bits, or 0/1
  int op = -1;         // gmlrt_sop, or -1 for an input register
  int input = -1;      // 0 face, 1 u, 2 v
  ExprP a, b, c;
  // linearisation state
  int uses = 0;
  int reg = -1;
  bool emitted = false;
  int64_t iv() const { return (int64_t)bits; }
  double rv() const { return vm_as_f64(bits); }
};

struct AClosure;
struct AVal;
using AVec = std::vector<AVal>;
struct AVal {
  enum K : uint8_t { Scalar, Point, Closure, Array, Other } k = Other;
  ExprP s;
  ExprP p[3];
  std::shared_ptr<const AClosure> clo;
  std::shared_ptr<const AVec> arr;  // storage order of the reference: reversed
  gml::ValuePtr other;
};
using AMap = std::unordered_map<std::string, AVal>;
struct AClosure {
  std::shared_ptr<const AMap> asnap;
  std::shared_ptr<const gml::EnvMap> base;  // concrete captured environment
  std::shared_ptr<const gml::AstList> ast;
};
using AStack = std::vector<AVal>;

struct Compiler {
  // hash-consing of expressions so that both arms of an `if` share common work
  std::map<std::tuple<int, int, const Expr*, const Expr*, const Expr*>, ExprP> cse;
  std::map<std::pair<int, uint64_t>, ExprP> const_cse;
  ExprP inputs[3];
  long steps = 0;
  // Run-time failures.  `path` is the condition under which the reference reaches the code being evaluated
  // (the conjunction of the enclosing data-dependent `if` conditions; null: always); an operation that can
  // fail at run time adds "path and it fails" to `traps`, and the program raises GMLRT_SOP_TRAP on their
  // disjunction.  Operations on the arm of an `if` that is not taken therefore raise nothing, although the
  // straight-line code computes both arms.
  ExprP path;
  std::vector<ExprP> traps;
  ExprP b_and(const ExprP& a, const ExprP& b) {
    if (!a) return b;
    if (!b) return a;
    if (a->is_const) return a->bits ? b : a;
    if (b->is_const) return b->bits ? a : b;
    return node(GMLRT_SOP_SELECT, Expr::B, a, b, kb(false));
  }
  ExprP b_or(const ExprP& a, const ExprP& b) {
    if (a->is_const) return a->bits ? a : b;
    if (b->is_const) return b->bits ? b : a;
    return node(GMLRT_SOP_SELECT, Expr::B, a, kb(true), b);
  }
  ExprP b_not(const ExprP& a) {
    if (a->is_const) return kb(!a->bits);
    return node(GMLRT_SOP_EQI, Expr::B, a, kb(false));
  }
  // `fails` (a boolean expression) holds where the operation being compiled fails in the reference
  void may_fail(const ExprP& fails, const char* what) {
    const ExprP cond = b_and(path, fails);
    if (cond->is_const) {
      if (cond->bits) throw gml::GmlError(what);  // fails whenever the function is called
      return;
    }
    traps.push_back(cond);
  }

  Compiler() {
    for (int i = 0; i < 3; i++) {
      inputs[i] = std::make_shared<Expr>();
      inputs[i]->ty = i == 0 ? Expr::I : Expr::R;
      inputs[i]->input = i;
    }
  }

  ExprP konst(Expr::Ty ty, uint64_t bits) {
    auto key = std::make_pair((int)ty, bits);
    auto it = const_cse.find(key);
    if (it != const_cse.end()) return it->second;
    auto e = std::make_shared<Expr>();
    e->ty = ty; e->is_const = true; e->bits = bits;
    const_cse[key] = e;
    return e;
  }
  ExprP ki(int64_t v) { return konst(Expr::I, (uint64_t)v); }
  ExprP kr(double v) { return konst(Expr::R, vm_from_f64(v)); }
  ExprP kb(bool v) { return konst(Expr::B, v ? 1 : 0); }

  ExprP node(int op, Expr::Ty ty, ExprP a, ExprP b = nullptr, ExprP c = nullptr) {
    auto key = std::make_tuple(op, (int)ty, (const Expr*)a.get(), (const Expr*)b.get(), (const Expr*)c.get());
    auto it = cse.find(key);
    if (it != cse.end()) return it->second;
    auto e = std::make_shared<Expr>();
    e->ty = ty; e->op = op; e->a = a; e->b = b; e->c = c;
    cse[key] = e;
    return e;
  }

  // ---- typed accessors
  static const ExprP& want(const AVal& v, Expr::Ty ty) {
    if (v.k != AVal::Scalar || v.s->ty != ty) throw gml::GmlError("WrongTypeRef");
    return v.s;
  }
  static AVal scalar(ExprP e) { AVal v; v.k = AVal::Scalar; v.s = std::move(e); return v; }

  AVal from_concrete(const gml::ValuePtr& v) {
    AVal r;
    switch (v->kind) {
      case gml::Value::Integer: return scalar(ki(v->i));
      case gml::Value::Real: return scalar(kr(v->r));
      case gml::Value::Boolean: return scalar(kb(v->b));
      case gml::Value::Point:
        r.k = AVal::Point;
        for (int i = 0; i < 3; i++) r.p[i] = kr(v->p[(size_t)i]);
        return r;
      case gml::Value::Closure: {
        auto c = std::make_shared<AClosure>();
        c->asnap = empty_amap();
        c->base = v->env;
        c->ast = v->ast;
        r.k = AVal::Closure; r.clo = c;
        return r;
      }
      case gml::Value::Array: {
        auto vec = std::make_shared<AVec>();
        for (auto& e : v->arr) vec->push_back(from_concrete(e));
        r.k = AVal::Array; r.arr = vec;
        return r;
      }
      default:
        r.k = AVal::Other; r.other = v;
        return r;
    }
  }

  static std::shared_ptr<const AMap> empty_amap() {
    static const std::shared_ptr<const AMap> e = std::make_shared<AMap>();
    return e;
  }

  // ---- environment with the reference's snapshot rule (evaluator.rs:87-100):
  // entries of the older snapshot win over the current bindings.
  struct AEnv {
    std::shared_ptr<const AMap> asnap;
    std::shared_ptr<const gml::EnvMap> base;
    AMap current;
  };
  bool lookup(const AEnv& e, const std::string& k, AVal* out) {
    auto ic = e.current.find(k);
    if (ic != e.current.end()) { *out = ic->second; return true; }
    auto is = e.asnap->find(k);
    if (is != e.asnap->end()) { *out = is->second; return true; }
    auto ib = e.base->find(k);
    if (ib != e.base->end()) { *out = from_concrete(ib->second); return true; }
    return false;
  }
  static std::shared_ptr<const AMap> snapshot(const AEnv& e) {
    if (e.current.empty()) return e.asnap;
    auto m = std::make_shared<AMap>(*e.asnap);
    for (auto& kv : e.current)
      if (!e.asnap->count(kv.first) && !e.base->count(kv.first)) (*m)[kv.first] = kv.second;
    return m;
  }

  // ---- scalar ops with constant folding through the evaluator's own arithmetic
  ExprP binop_i(int op, const ExprP& a, const ExprP& b) {
    if (a->is_const && b->is_const) {
      const int64_t x = a->iv(), y = b->iv();
      switch (op) {
        case GMLRT_SOP_ADDI: return ki((int64_t)((uint64_t)x + (uint64_t)y));
        case GMLRT_SOP_SUBI: return ki((int64_t)((uint64_t)x - (uint64_t)y));
        case GMLRT_SOP_MULI: return ki((int64_t)((uint64_t)x * (uint64_t)y));
        case GMLRT_SOP_DIVI: return ki(gml::divi(x, y));
        case GMLRT_SOP_MODI: return ki(gml::modi(x, y));
        case GMLRT_SOP_EQI: return kb(x == y);
        case GMLRT_SOP_LESSI: return kb(x < y);
      }
    }
    const bool pred = op == GMLRT_SOP_EQI || op == GMLRT_SOP_LESSI;
    return node(op, pred ? Expr::B : Expr::I, a, b);
  }
  ExprP binop_r(int op, const ExprP& a, const ExprP& b) {
    if (a->is_const && b->is_const) {
      const double x = a->rv(), y = b->rv();
      switch (op) {
        case GMLRT_SOP_ADDF: return kr(x + y);
        case GMLRT_SOP_SUBF: return kr(x - y);
        case GMLRT_SOP_MULF: return kr(x * y);
        case GMLRT_SOP_DIVF: return kr(x / y);
        case GMLRT_SOP_EQF: return kb(x == y);
        case GMLRT_SOP_LESSF: return kb(x < y);
      }
    }
    const bool pred = op == GMLRT_SOP_EQF || op == GMLRT_SOP_LESSF;
    return node(op, pred ? Expr::B : Expr::R, a, b);
  }
  ExprP unop_r(int op, const ExprP& a) {
    if (a->is_const) {
      const double x = a->rv();
      switch (op) {
        case GMLRT_SOP_SQRT: return kr(std::sqrt(x));
        case GMLRT_SOP_FRAC: return kr(gml::gml_frac(x));
        case GMLRT_SOP_CLAMPF: return kr(gml::gml_clampf(x));
        case GMLRT_SOP_SIN: return kr(gml::gml_sin(x));
        case GMLRT_SOP_COS: return kr(gml::gml_cos(x));
        case GMLRT_SOP_ASIN: return kr(gml::gml_asin(x));
        case GMLRT_SOP_ACOS: return kr(gml::gml_acos(x));
      }
    }
    return node(op, Expr::R, a);
  }

  ExprP select_expr(const ExprP& pred, const ExprP& x, const ExprP& y) {
    if (x.get() == y.get()) return x;
    if (x->ty != y->ty) throw UnsupportedSurface("`if` arms leave values of different types");
    return node(GMLRT_SOP_SELECT, x->ty, pred, x, y);
  }
  AVal select_val(const ExprP& pred, const AVal& x, const AVal& y) {
    if (x.k != y.k) throw UnsupportedSurface("`if` arms leave values of different kinds");
    switch (x.k) {
      case AVal::Scalar: return scalar(select_expr(pred, x.s, y.s));
      case AVal::Point: {
        AVal r; r.k = AVal::Point;
        for (int i = 0; i < 3; i++) r.p[i] = select_expr(pred, x.p[i], y.p[i]);
        return r;
      }
      case AVal::Closure:
        if (x.clo.get() == y.clo.get()) return x;
        throw UnsupportedSurface("`if` arms leave different closures");
      case AVal::Array:
        if (x.arr.get() == y.arr.get()) return x;
        if (x.arr->size() == y.arr->size()) {
          auto vec = std::make_shared<AVec>();
          for (size_t i = 0; i < x.arr->size(); i++) vec->push_back(select_val(pred, (*x.arr)[i], (*y.arr)[i]));
          AVal r; r.k = AVal::Array; r.arr = vec;
          return r;
        }
        throw UnsupportedSurface("`if` arms leave arrays of different length");
      default:
        if (x.other.get() == y.other.get()) return x;
        throw UnsupportedSurface("`if` arms leave different non-scalar values");
    }
  }

  static AVal pop(AStack& s) {
    if (s.empty()) throw gml::GmlError("EmptyStack");
    AVal v = std::move(s.back());
    s.pop_back();
    return v;
  }

  void call(const AVal& c, AStack& stack) {
    if (c.k != AVal::Closure) throw gml::GmlError("WrongType");
    // loops are recursion in GML; one that does not end on constants would unroll for ever
    if (++depth > 2000) throw UnsupportedSurface("surface function recurses on a data-dependent condition");
    AEnv e{c.clo->asnap, c.clo->base, {}};
    eval(e, stack, *c.clo->ast);
    --depth;
  }
  int depth = 0;

  void eval_op(gml::Op op, AStack& st) {
    using gml::Op;
    auto bin_i = [&](int sop) { AVal b = pop(st), a = pop(st); st.push_back(scalar(binop_i(sop, want(a, Expr::I), want(b, Expr::I)))); };
    auto bin_r = [&](int sop) { AVal b = pop(st), a = pop(st); st.push_back(scalar(binop_r(sop, want(a, Expr::R), want(b, Expr::R)))); };
    auto un_r = [&](int sop) { AVal a = pop(st); st.push_back(scalar(unop_r(sop, want(a, Expr::R)))); };
    switch (op) {
      case Op::Addi: bin_i(GMLRT_SOP_ADDI); break;
      case Op::Subi: bin_i(GMLRT_SOP_SUBI); break;
      case Op::Muli: bin_i(GMLRT_SOP_MULI); break;
      case Op::Divi: case Op::Modi: {
        // evaluator.rs:357, 422: a zero divisor panics
        AVal b = pop(st), a = pop(st);
        const ExprP &ea = want(a, Expr::I), &eb = want(b, Expr::I);
        const int sop = op == Op::Divi ? GMLRT_SOP_DIVI : GMLRT_SOP_MODI;
        if (eb->is_const && eb->iv() == 0) {
          may_fail(kb(true), "attempt to divide by zero");
          st.push_back(scalar(ki(0)));  // what the bytecode defines for the arm that must not be taken
        } else {
          if (!eb->is_const) may_fail(binop_i(GMLRT_SOP_EQI, eb, ki(0)), "attempt to divide by zero");
          st.push_back(scalar(binop_i(sop, ea, eb)));
        }
        break;
      }
      case Op::Eqi: bin_i(GMLRT_SOP_EQI); break;
      case Op::Lessi: bin_i(GMLRT_SOP_LESSI); break;
      case Op::Addf: bin_r(GMLRT_SOP_ADDF); break;
      case Op::Subf: bin_r(GMLRT_SOP_SUBF); break;
      case Op::Mulf: bin_r(GMLRT_SOP_MULF); break;
      case Op::Divf: bin_r(GMLRT_SOP_DIVF); break;
      case Op::Eqf: bin_r(GMLRT_SOP_EQF); break;
      case Op::Lessf: bin_r(GMLRT_SOP_LESSF); break;
      case Op::Sqrt: un_r(GMLRT_SOP_SQRT); break;
      case Op::Frac: un_r(GMLRT_SOP_FRAC); break;
      case Op::Clampf: un_r(GMLRT_SOP_CLAMPF); break;
      case Op::Sin: un_r(GMLRT_SOP_SIN); break;
      case Op::Cos: un_r(GMLRT_SOP_COS); break;
      case Op::Asin: un_r(GMLRT_SOP_ASIN); break;
      case Op::Acos: un_r(GMLRT_SOP_ACOS); break;
      case Op::Negi: { AVal a = pop(st); st.push_back(scalar(want(a, Expr::I))); break; }  // no-op in the reference
      case Op::Negf: { AVal a = pop(st); st.push_back(scalar(want(a, Expr::R))); break; }
      case Op::Real: {
        AVal a = pop(st);
        const ExprP& e = want(a, Expr::I);
        st.push_back(scalar(e->is_const ? kr((double)e->iv()) : node(GMLRT_SOP_REAL, Expr::R, e)));
        break;
      }
      case Op::Floor: {
        AVal a = pop(st);
        const ExprP& e = want(a, Expr::R);
        st.push_back(scalar(e->is_const ? ki(gml::gml_floor(e->rv())) : node(GMLRT_SOP_FLOOR, Expr::I, e)));
        break;
      }
      case Op::Point: {
        AVal z = pop(st), y = pop(st), x = pop(st);
        AVal r; r.k = AVal::Point;
        r.p[0] = want(x, Expr::R); r.p[1] = want(y, Expr::R); r.p[2] = want(z, Expr::R);
        st.push_back(r);
        break;
      }
      case Op::Getx: case Op::Gety: case Op::Getz: {
        AVal p = pop(st);
        if (p.k != AVal::Point) throw gml::GmlError("WrongTypeRef");
        st.push_back(scalar(p.p[op == Op::Getx ? 0 : op == Op::Gety ? 1 : 2]));
        break;
      }
      case Op::Length: {
        AVal a = pop(st);
        if (a.k != AVal::Array) throw gml::GmlError("WrongTypeRef");
        st.push_back(scalar(ki((int64_t)a.arr->size())));
        break;
      }
      case Op::Get: {
        AVal i = pop(st), a = pop(st);
        const ExprP& ie = want(i, Expr::I);
        if (a.k != AVal::Array) throw gml::GmlError("WrongTypeRef");
        const int64_t len = (int64_t)a.arr->size();
        if (ie->is_const) {
          const int64_t iv = ie->iv();
          if (iv < 0 || iv >= len) {
            may_fail(kb(true), "ArrayOutOfBounds");
            if (len == 0) throw UnsupportedSurface("`get` on an empty array on a conditional path");
            st.push_back((*a.arr)[(size_t)(len - 1)]);
            break;
          }
          st.push_back((*a.arr)[(size_t)(len - iv - 1)]);
        } else {
          // data-dependent index: a chain of selects over the elements.  An index outside
          // [0,len) aborts the reference (evaluator.rs:544-555); here it yields element 0 and raises the trap.
          if (len == 0) { may_fail(kb(true), "ArrayOutOfBounds"); throw UnsupportedSurface("`get` on an empty array on a conditional path"); }
          may_fail(b_or(binop_i(GMLRT_SOP_LESSI, ie, ki(0)), binop_i(GMLRT_SOP_LESSI, ki(len - 1), ie)), "ArrayOutOfBounds");
          if (len > 256) throw UnsupportedSurface("data-dependent `get` on an array longer than 256");
          AVal r = (*a.arr)[(size_t)(len - 1)];
          for (int64_t k = 1; k < len; k++)
            r = select_val(binop_i(GMLRT_SOP_EQI, ie, ki(k)), (*a.arr)[(size_t)(len - k - 1)], r);
          st.push_back(r);
        }
        break;
      }
      case Op::Apply: { AVal c = pop(st); call(c, st); break; }
      case Op::If: {
        AVal c2 = pop(st), c1 = pop(st), pred = pop(st);
        const ExprP& pe = want(pred, Expr::B);
        if (pe->is_const) {
          call(pe->bits ? c1 : c2, st);
        } else {
          AStack s1 = st, s2 = st;
          const ExprP outer = path;
          path = b_and(outer, pe);
          call(c1, s1);
          path = b_and(outer, b_not(pe));
          call(c2, s2);
          path = outer;
          if (s1.size() != s2.size()) throw UnsupportedSurface("`if` arms leave stacks of different depth");
          st.clear();
          for (size_t k = 0; k < s1.size(); k++) st.push_back(select_val(pe, s1[k], s2[k]));
        }
        break;
      }
      default:
        throw UnsupportedSurface(std::string("operator `") + gml::op_name(op) + "` inside a surface function");
    }
  }

  void eval(AEnv& env, AStack& st, const gml::AstList& ast) {
    for (const gml::AstNode& item : ast) {
      if (++steps > 2000000) throw UnsupportedSurface("surface function does not terminate under symbolic evaluation");
      switch (item.kind) {
        case gml::AstNode::Function: {
          auto c = std::make_shared<AClosure>();
          c->asnap = snapshot(env);
          c->base = env.base;
          c->ast = item.children;
          AVal v; v.k = AVal::Closure; v.clo = c;
          st.push_back(v);
          break;
        }
        case gml::AstNode::Array: {
          AStack local;
          eval(env, local, *item.children);
          auto vec = std::make_shared<AVec>(local.rbegin(), local.rend());
          AVal v; v.k = AVal::Array; v.arr = vec;
          st.push_back(v);
          break;
        }
        case gml::AstNode::Leaf: {
          const gml::Token& t = item.tok;
          switch (t.kind) {
            case gml::Token::Integer: st.push_back(scalar(ki(t.i))); break;
            case gml::Token::Real: st.push_back(scalar(kr(t.r))); break;
            case gml::Token::Boolean: st.push_back(scalar(kb(t.b))); break;
            case gml::Token::Str: {
              auto sv = std::make_shared<gml::Value>();
              sv->kind = gml::Value::String; sv->s = t.s;
              AVal v; v.k = AVal::Other; v.other = sv;
              st.push_back(v);
              break;
            }
            case gml::Token::Binder: env.current[t.s] = pop(st); break;
            case gml::Token::Identifier: {
              AVal v;
              if (!lookup(env, t.s, &v)) throw gml::GmlError("environment missing identifier: " + t.s);
              st.push_back(v);
              break;
            }
            case gml::Token::Operator: eval_op(t.op, st); break;
            default: throw gml::GmlError("InvalidAst");
          }
          break;
        }
      }
    }
  }

  // ---- linearisation
  struct Emit {
    std::vector<uint64_t> code;
    std::vector<uint64_t> consts;
    std::map<uint64_t, uint32_t> const_idx;
    std::vector<int> free_regs;
    int next_reg = 3;
  };

  static void count_uses(const ExprP& e) {
    if (!e || e->is_const) return;
    if (e->uses++ > 0) return;
    count_uses(e->a); count_uses(e->b); count_uses(e->c);
  }
  static uint32_t operand(Emit& em, const ExprP& e) {
    if (!e) return 0;
    if (e->is_const) {
      auto it = em.const_idx.find(e->bits);
      uint32_t idx;
      if (it == em.const_idx.end()) {
        idx = (uint32_t)em.consts.size();
        if (idx >= 0x8000u) throw UnsupportedSurface("surface function needs more than 32768 constants");
        em.consts.push_back(e->bits);
        em.const_idx[e->bits] = idx;
      } else {
        idx = it->second;
      }
      return 0x8000u | idx;
    }
    return (uint32_t)e->reg;
  }
  static void release(Emit& em, const ExprP& e) {
    if (!e || e->is_const || e->input >= 0) return;
    if (--e->uses == 0) em.free_regs.push_back(e->reg);
  }
  static void emit_expr(Emit& em, const ExprP& e) {
    if (!e || e->is_const || e->input >= 0 || e->emitted) return;
    emit_expr(em, e->a); emit_expr(em, e->b); emit_expr(em, e->c);
    const uint32_t oa = operand(em, e->a), ob = operand(em, e->b), oc = operand(em, e->c);
    // an operand dies here once per reference to it
    release(em, e->a); release(em, e->b); release(em, e->c);
    int reg;
    if (!em.free_regs.empty()) { reg = em.free_regs.back(); em.free_regs.pop_back(); }
    else reg = em.next_reg++;
    if (reg >= GMLRT_SURF_MAX_REGS) throw UnsupportedSurface("surface function needs more than 64 live values");
    e->reg = reg;
    e->emitted = true;
    em.code.push_back((uint64_t)(uint32_t)e->op | ((uint64_t)(uint32_t)reg << 8) | ((uint64_t)oa << 16) |
                      ((uint64_t)ob << 32) | ((uint64_t)oc << 48));
  }
};

}  // namespace

SurfaceProgram compile_surface(const gml::Value& closure) {
  if (closure.kind != gml::Value::Closure) throw gml::GmlError("WrongTypeRef");
  Compiler c;
  for (int i = 0; i < 3; i++) c.inputs[i]->reg = i;
  Compiler::AEnv env{Compiler::empty_amap(), closure.env, {}};
  AStack st;
  for (int i = 0; i < 3; i++) st.push_back(Compiler::scalar(c.inputs[i]));
  c.eval(env, st, *closure.ast);
  // pop n, ks, kd, colour (evaluator.rs:784-791)
  AVal n = Compiler::pop(st), ks = Compiler::pop(st), kd = Compiler::pop(st), col = Compiler::pop(st);
  ExprP out[6];
  if (col.k != AVal::Point) throw gml::GmlError("wrong type");
  for (int i = 0; i < 3; i++) out[i] = col.p[i];
  out[3] = Compiler::want(kd, Expr::R);
  out[4] = Compiler::want(ks, Expr::R);
  out[5] = Compiler::want(n, Expr::R);

  ExprP trap;  // disjunction of the failure conditions collected on the way
  for (auto& t : c.traps) trap = trap ? c.b_or(trap, t) : t;
  SurfaceProgram sp;
  bool all_const = !trap;
  for (auto& e : out) all_const = all_const && e->is_const;
  if (all_const) {
    sp.constant = true;
    for (auto& e : out) sp.consts.push_back(e->bits);
    return sp;
  }
  Compiler::Emit em;
  // each OUT operand counts as a use so that its register stays live until the OUT
  for (auto& e : out) Compiler::count_uses(e);
  if (trap) Compiler::count_uses(trap);
  for (auto& e : out) Compiler::emit_expr(em, e);
  if (trap) {
    Compiler::emit_expr(em, trap);
    em.code.push_back((uint64_t)GMLRT_SOP_TRAP | ((uint64_t)Compiler::operand(em, trap) << 16));
    Compiler::release(em, trap);
  }
  for (int g = 0; g < 2; g++) {
    const uint32_t oa = Compiler::operand(em, out[g * 3]), ob = Compiler::operand(em, out[g * 3 + 1]),
                   oc = Compiler::operand(em, out[g * 3 + 2]);
    em.code.push_back((uint64_t)GMLRT_SOP_OUT | ((uint64_t)(g * 3) << 8) | ((uint64_t)oa << 16) | ((uint64_t)ob << 32) |
                      ((uint64_t)oc << 48));
  }
  em.code.push_back(GMLRT_SOP_END);
  sp.code = std::move(em.code);
  sp.consts = std::move(em.consts);
  return sp;
}

void run_surface_words(const SurfaceProgram& p, int64_t face, double u, double v, double out[6], bool* trapped) {
  if (trapped) *trapped = false;
  if (p.constant) {
    for (int i = 0; i < 6; i++) out[i] = vm_as_f64(p.consts[(size_t)i]);
    return;
  }
  uint64_t regs[GMLRT_SURF_MAX_REGS] = {0};
  const VmRegs<uint64_t*, uint64_t*> R = {regs, 1, GMLRT_SURF_MAX_REGS, regs, 1, true, GMLRT_SURF_MAX_REGS, 0};
  surface_vm<true>(p.code.data(), p.consts.data(), face, u, v, out, R, trapped);
}

void expand_surface(const uint64_t* code, const uint64_t* consts, std::vector<uint8_t>& out) {
  for (size_t pc = 0;; pc++) {
    const uint64_t w = code[pc];
    SurfInstr I;
    std::memset(&I, 0, sizeof(I));
    const uint32_t op = (uint32_t)(w & 0xFF);
    I.head = (uint32_t)(w & 0xFFFF);
    const uint32_t x[3] = {(uint32_t)((w >> 16) & 0xFFFF), (uint32_t)((w >> 32) & 0xFFFF), (uint32_t)((w >> 48) & 0xFFFF)};
    uint64_t* slot[3] = {&I.a, &I.b, &I.c};
    const uint32_t imm[3] = {GMLRT_SI_IMM_A, GMLRT_SI_IMM_B, GMLRT_SI_IMM_C};
    if (op != GMLRT_SOP_END)
      for (int k = 0; k < 3; k++) {
        if (x[k] & 0x8000u) { *slot[k] = consts[x[k] & 0x7FFFu]; I.head |= imm[k]; }
        else *slot[k] = x[k] & (GMLRT_SURF_MAX_REGS - 1);
      }
    const uint8_t* b = (const uint8_t*)&I;
    out.insert(out.end(), b, b + sizeof(I));
    if (op == GMLRT_SOP_END) break;
  }
}

void run_surface_program(const SurfaceProgram& p, int64_t face, double u, double v, double out[6], bool* trapped) {
  if (trapped) *trapped = false;
  if (p.constant) {
    for (int i = 0; i < 6; i++) out[i] = vm_as_f64(p.consts[(size_t)i]);
    return;
  }
  std::vector<uint8_t> rec;
  expand_surface(p.code.data(), p.consts.data(), rec);
  uint64_t regs[GMLRT_SURF_MAX_REGS] = {0};
  const VmRegs<uint64_t*, uint64_t*> R = {regs, 1, GMLRT_SURF_MAX_REGS, regs, 1, true, GMLRT_SURF_MAX_REGS, 0};
  surface_stream_vm<true>((const SurfInstr*)rec.data(), face, u, v, VmArrayOut{out, trapped}, R);
}

// ---- specialisation of an expanded program for one value of `face`.
// The face index is the one input of a surface function with a handful of values, and scene
// files branch on it at the top (`face 0 eqi { side } { caps } if`): both arms are in the
// program, joined by SELECT.  With face known the comparison folds, the SELECT turns into a
// move and the arm not taken is dead code.  Folding uses the interpreter's own scalar
// functions and leaves out the libm-class operations (sin, cos, asin, acos), whose last bit
// may differ between host and device; everything else is IEEE-exact on both.
namespace {
// which of a, b, c an operation reads
void surf_reads(uint32_t op, bool r[3]) {
  r[0] = r[1] = r[2] = false;
  switch (op) {
    case GMLRT_SOP_END: return;
    case GMLRT_SOP_MOV: case GMLRT_SOP_REAL: case GMLRT_SOP_FLOOR: case GMLRT_SOP_SQRT: case GMLRT_SOP_FRAC:
    case GMLRT_SOP_CLAMPF: case GMLRT_SOP_SIN: case GMLRT_SOP_COS: case GMLRT_SOP_ASIN: case GMLRT_SOP_ACOS:
    case GMLRT_SOP_TRAP:
      r[0] = true; return;
    case GMLRT_SOP_SELECT: case GMLRT_SOP_OUT: r[0] = r[1] = r[2] = true; return;
    default: r[0] = r[1] = true; return;
  }
}
bool surf_fold(uint32_t op, uint64_t a, uint64_t b, uint64_t* res) {
  const double fa = vm_as_f64(a), fb = vm_as_f64(b);
  const int64_t ia = (int64_t)a, ib = (int64_t)b;
  switch (op) {
    case GMLRT_SOP_MOV: *res = a; return true;
    case GMLRT_SOP_ADDI: *res = a + b; return true;
    case GMLRT_SOP_SUBI: *res = a - b; return true;
    case GMLRT_SOP_MULI: *res = a * b; return true;
    case GMLRT_SOP_DIVI: *res = (uint64_t)vm_divi(ia, ib); return true;
    case GMLRT_SOP_MODI: *res = (uint64_t)vm_modi(ia, ib); return true;
    case GMLRT_SOP_EQI: *res = ia == ib; return true;
    case GMLRT_SOP_LESSI: *res = ia < ib; return true;
    case GMLRT_SOP_REAL: *res = vm_from_f64((double)ia); return true;
    case GMLRT_SOP_FLOOR: *res = (uint64_t)vm_floor_i64(fa); return true;
    case GMLRT_SOP_ADDF: *res = vm_from_f64(fa + fb); return true;
    case GMLRT_SOP_SUBF: *res = vm_from_f64(fa - fb); return true;
    case GMLRT_SOP_MULF: *res = vm_from_f64(fa * fb); return true;
    case GMLRT_SOP_DIVF: *res = vm_from_f64(fa / fb); return true;
    case GMLRT_SOP_EQF: *res = fa == fb; return true;
    case GMLRT_SOP_LESSF: *res = fa < fb; return true;
    case GMLRT_SOP_SQRT: *res = vm_from_f64(sqrt(fa)); return true;
    case GMLRT_SOP_FRAC: *res = vm_from_f64(vm_frac(fa)); return true;
    case GMLRT_SOP_CLAMPF: *res = vm_from_f64(vm_clampf(fa)); return true;
    default: return false;
  }
}
}  // namespace

void specialize_surface(const uint8_t* records, int64_t face, std::vector<uint8_t>& out) {
  const SurfInstr* prog = (const SurfInstr*)records;
  const uint32_t imm[3] = {GMLRT_SI_IMM_A, GMLRT_SI_IMM_B, GMLRT_SI_IMM_C};
  // forward: registers with a known value become immediates, operations on immediates fold
  bool known[GMLRT_SURF_MAX_REGS] = {false};
  uint64_t value[GMLRT_SURF_MAX_REGS] = {0};
  known[0] = true;
  value[0] = (uint64_t)face;
  // copies: alias[r] = s while register r holds what a move took from register s and neither has been
  // written since; readers of r are pointed at s, which leaves the move without readers (dropped below)
  int alias[GMLRT_SURF_MAX_REGS];
  for (int& a : alias) a = -1;
  auto written = [&](uint32_t r) {
    alias[r] = -1;
    for (int& a : alias) if (a == (int)r) a = -1;
  };
  std::vector<SurfInstr> fwd;
  for (size_t pc = 0;; pc++) {
    SurfInstr I = prog[pc];
    const uint32_t op = I.head & 0xFFu;
    if (op == GMLRT_SOP_END) { fwd.push_back(I); break; }
    bool reads[3];
    surf_reads(op, reads);
    uint64_t* slot[3] = {&I.a, &I.b, &I.c};
    for (int k = 0; k < 3; k++) {
      if (!reads[k]) { *slot[k] = 0; I.head |= imm[k]; continue; }  // an operand nobody looks at
      if (!(I.head & imm[k]) && known[*slot[k] & (GMLRT_SURF_MAX_REGS - 1)]) {
        *slot[k] = value[*slot[k] & (GMLRT_SURF_MAX_REGS - 1)];
        I.head |= imm[k];
      } else if (!(I.head & imm[k]) && alias[*slot[k] & (GMLRT_SURF_MAX_REGS - 1)] >= 0) {
        *slot[k] = (uint64_t)alias[*slot[k] & (GMLRT_SURF_MAX_REGS - 1)];
      }
    }
    if (op == GMLRT_SOP_OUT) { fwd.push_back(I); continue; }
    if (op == GMLRT_SOP_TRAP) {
      if (!((I.head & imm[0]) && I.a == 0)) fwd.push_back(I);  // a failure that cannot happen on this face is dropped
      continue;
    }
    const uint32_t dst = (I.head >> 8) & (GMLRT_SURF_MAX_REGS - 1);
    if (op == GMLRT_SOP_SELECT) {
      const bool same = ((I.head & imm[1]) != 0) == ((I.head & imm[2]) != 0) && I.b == I.c;
      if ((I.head & imm[0]) || same) {
        // a move of the chosen side
        const bool take_b = same || I.a != 0;
        const uint64_t v = take_b ? I.b : I.c;
        const bool v_imm = (I.head & (take_b ? imm[1] : imm[2])) != 0;
        if (v_imm) { known[dst] = true; value[dst] = v; written(dst); continue; }
        if ((v & (GMLRT_SURF_MAX_REGS - 1)) == dst) continue;  // onto itself: the value is unknown as before
        SurfInstr M;
        std::memset(&M, 0, sizeof(M));
        M.head = (uint32_t)GMLRT_SOP_MOV | (dst << 8) | imm[1] | imm[2];
        M.a = v;
        fwd.push_back(M);
        known[dst] = false;
        written(dst);
        alias[dst] = (int)(v & (GMLRT_SURF_MAX_REGS - 1));
        continue;
      }
    } else {
      bool all_imm = true;
      for (int k = 0; k < 3; k++) all_imm = all_imm && (I.head & imm[k]);
      uint64_t res;
      if (all_imm && surf_fold(op, I.a, I.b, &res)) { known[dst] = true; value[dst] = res; written(dst); continue; }
    }
    known[dst] = false;
    written(dst);
    if (op == GMLRT_SOP_MOV && !(I.head & imm[0])) {
      if ((I.a & (GMLRT_SURF_MAX_REGS - 1)) == dst) continue;  // a move onto itself
      alias[dst] = (int)(I.a & (GMLRT_SURF_MAX_REGS - 1));
    }
    fwd.push_back(I);
  }
  // backward: drop what no OUT depends on
  bool live[GMLRT_SURF_MAX_REGS] = {false};
  std::vector<SurfInstr> kept;
  for (size_t i = fwd.size(); i-- > 0;) {
    const SurfInstr& I = fwd[i];
    const uint32_t op = I.head & 0xFFu;
    if (op != GMLRT_SOP_END && op != GMLRT_SOP_OUT && op != GMLRT_SOP_TRAP) {
      const uint32_t dst = (I.head >> 8) & (GMLRT_SURF_MAX_REGS - 1);
      if (!live[dst]) continue;
      live[dst] = false;
    }
    const uint64_t opnd[3] = {I.a, I.b, I.c};
    for (int k = 0; k < 3; k++)
      if (!(I.head & imm[k])) live[opnd[k] & (GMLRT_SURF_MAX_REGS - 1)] = true;
    kept.push_back(I);
  }
  for (size_t i = kept.size(); i-- > 0;) {
    const uint8_t* b = (const uint8_t*)&kept[i];
    out.insert(out.end(), b, b + sizeof(SurfInstr));
  }
}

void run_surface_specialized(const SurfaceProgram& p, int64_t face, double u, double v, double out[6], bool* trapped) {
  if (trapped) *trapped = false;
  if (p.constant) {
    for (int i = 0; i < 6; i++) out[i] = vm_as_f64(p.consts[(size_t)i]);
    return;
  }
  std::vector<uint8_t> rec, spec;
  expand_surface(p.code.data(), p.consts.data(), rec);
  specialize_surface(rec.data(), face, spec);
  uint64_t regs[GMLRT_SURF_MAX_REGS] = {0};
  const VmRegs<uint64_t*, uint64_t*> R = {regs, 1, GMLRT_SURF_MAX_REGS, regs, 1, true, GMLRT_SURF_MAX_REGS, 0};
  surface_stream_vm<true>((const SurfInstr*)spec.data(), face, u, v, VmArrayOut{out, trapped}, R);
}

// ============================================================== flattening
namespace {

constexpr double kInf = std::numeric_limits<double>::infinity();

struct Flattener : gml::TreeVisitor {
  std::vector<gmlrt_leaf> leaves;
  std::vector<uint32_t> nodes;
  std::vector<uint64_t> code;
  std::vector<uint64_t> consts;
  std::map<const gml::Value*, std::pair<uint32_t, uint32_t>> by_closure;
  std::map<std::vector<uint64_t>, uint32_t> by_code, by_consts;
  // operand stack of the postfix program: (is_leaf, worst-case event count)
  std::vector<std::pair<bool, uint64_t>> ostack;
  uint64_t sp = 0, max_events = 0;
  size_t max_stack = 0;

  void leaf(gml::PrimType prim, const gml::ValuePtr& surface, const gml::Transform& t) override {
    gmlrt_leaf L;
    std::memset(&L, 0, sizeof(L));
    for (int i = 0; i < 3; i++)
      for (int j = 0; j < 4; j++) {
        L.inv_m[i * 4 + j] = t.inv_m.m[i][j];
        L.m[i * 4 + j] = t.m.m[i][j];
      }
    L.type = (uint32_t)prim;
    auto it = by_closure.find(surface.get());
    if (it == by_closure.end()) {
      SurfaceProgram sp = compile_surface(*surface);
      uint32_t code_off = GMLRT_SURF_CONST;
      if (!sp.constant) {
        auto ic = by_code.find(sp.code);
        if (ic == by_code.end()) {
          code_off = (uint32_t)code.size();
          code.insert(code.end(), sp.code.begin(), sp.code.end());
          by_code[sp.code] = code_off;
        } else {
          code_off = ic->second;
        }
      }
      uint32_t const_off;
      auto ik = by_consts.find(sp.consts);
      if (ik == by_consts.end()) {
        const_off = (uint32_t)consts.size();
        consts.insert(consts.end(), sp.consts.begin(), sp.consts.end());
        by_consts[sp.consts] = const_off;
      } else {
        const_off = ik->second;
      }
      it = by_closure.emplace(surface.get(), std::make_pair(code_off, const_off)).first;
    }
    L.surf_code = it->second.first;
    L.surf_const = it->second.second;
    if (leaves.size() >= (1u << 20)) throw gml::GmlError("scene has more than 2^20 leaves");
    FNode n;
    n.leaf = (int)leaves.size();
    n.n_leaves = 1;
    n.bnd = leaf_bound(prim, t);
    leaves.push_back(L);
    tstack.push_back((int)tree.size());
    tree.push_back(n);
  }
  void op(gml::Rule rule) override {
    FNode n;
    n.b = tstack.back(); tstack.pop_back();
    n.a = tstack.back(); tstack.pop_back();
    n.rule = rule;
    n.n_leaves = tree[(size_t)n.a].n_leaves + tree[(size_t)n.b].n_leaves;
    const Bound &ba = tree[(size_t)n.a].bnd, &bb = tree[(size_t)n.b].bnd;
    // a point of the result lies in A or B (union), in A and B (intersect), in A (difference)
    if (rule == gml::Rule::Union) {
      n.bnd = sphere_union(ba, bb);
      for (int i = 0; i < 3; i++) { n.bnd.lo[i] = std::min(ba.lo[i], bb.lo[i]); n.bnd.hi[i] = std::max(ba.hi[i], bb.hi[i]); }
    } else if (rule == gml::Rule::Intersect) {
      n.bnd = (!bb.finite || (ba.finite && ba.r <= bb.r)) ? ba : bb;
      for (int i = 0; i < 3; i++) { n.bnd.lo[i] = std::max(ba.lo[i], bb.lo[i]); n.bnd.hi[i] = std::min(ba.hi[i], bb.hi[i]); }
    } else {
      n.bnd = ba;
    }
    tighten(n.bnd);
    tstack.push_back((int)tree.size());
    tree.push_back(n);
  }

  // ---- bounding volumes (acceleration hints; see GMLRT_NODE_BOUND in include/gmlrt.h):
  // a sphere (finite = false: none known) and an axis-aligned box (sides may be infinite)
  struct Bound {
    bool finite = false;
    double c[3] = {0, 0, 0};
    double r = 0;
    double lo[3] = {-kInf, -kInf, -kInf}, hi[3] = {kInf, kInf, kInf};
    bool box_finite() const {
      for (int i = 0; i < 3; i++) if (!std::isfinite(lo[i]) || !std::isfinite(hi[i])) return false;
      return true;
    }
    bool box_empty() const { return lo[0] > hi[0] || lo[1] > hi[1] || lo[2] > hi[2]; }
    // Surface area of the tighter of the two volumes: for convex bodies the chance that a
    // random ray meets one is proportional to it.
    double area() const {
      double a = finite ? 4.0 * 3.14159265358979323846 * r * r : kInf;
      if (box_empty()) return 0.0;
      if (box_finite()) {
        const double e[3] = {hi[0] - lo[0], hi[1] - lo[1], hi[2] - lo[2]};
        a = std::min(a, 2.0 * (e[0] * e[1] + e[1] * e[2] + e[2] * e[0]));
      }
      return a;
    }
  };
  // Each volume bounds the other: a finite box has a circumscribed sphere, a sphere a
  // circumscribed box.
  static void tighten(Bound& b) {
    if (b.box_empty()) {
      // an empty solid: any small finite volume will do
      if (!b.finite) { b.finite = true; b.c[0] = b.c[1] = b.c[2] = 0.0; }
      b.r = 0.0;
      return;
    }
    if (b.box_finite()) {
      double h2 = 0.0, bc[3];
      for (int i = 0; i < 3; i++) { bc[i] = 0.5 * (b.lo[i] + b.hi[i]); const double h = 0.5 * (b.hi[i] - b.lo[i]); h2 += h * h; }
      const double br = std::sqrt(h2) * (1.0 + 1e-9) + 1e-12;
      if (std::isfinite(br) && (!b.finite || br < b.r)) { b.finite = true; b.r = br; for (int i = 0; i < 3; i++) b.c[i] = bc[i]; }
    }
    if (b.finite)
      for (int i = 0; i < 3; i++) {
        b.lo[i] = std::max(b.lo[i], b.c[i] - b.r - 1e-12 * std::fabs(b.c[i]));
        b.hi[i] = std::min(b.hi[i], b.c[i] + b.r + 1e-12 * std::fabs(b.c[i]));
      }
  }
  struct FNode { int a = -1, b = -1, leaf = -1; gml::Rule rule = gml::Rule::Union; uint32_t n_leaves = 0; Bound bnd; };
  std::vector<FNode> tree;
  std::vector<int> tstack;
  std::vector<gmlrt_bound> bounds;

  static Bound leaf_bound(gml::PrimType prim, const gml::Transform& t) {
    Bound b;
    // the primitive's box in object space (primitives.rs: sphere |p| <= 1; cube [0,1]^3;
    // cylinder and cone radius 1, 0 <= y <= 1)
    double lo[3] = {-1, -1, -1}, hi[3] = {1, 1, 1};
    switch (prim) {
      case gml::PrimType::Sphere: break;
      case gml::PrimType::Cube: lo[0] = lo[1] = lo[2] = 0.0; break;
      case gml::PrimType::Cylinder:
        // The reference's cylinder reports hits slightly OUTSIDE the solid for rays within 1e-7 of parallel or
        // orthogonal to the axis (primitives.rs:644-663 test only the ray's origin against the radius, or against
        // the two caps, and let the ray drift from there).  The box is widened by a thousandth of the unit
        // cylinder so that such a hit still lies inside it: that covers an orthogonal ray's drift along the axis
        // over 10^4 object units and a parallel ray's sideways drift (4.5e-4 per unit) over two.
        lo[0] = lo[2] = -1.001; hi[0] = hi[2] = 1.001; lo[1] = -0.001; hi[1] = 1.001;
        break;
      case gml::PrimType::Cone:
        // The small root of the reference's quadratic, (-b + sqrt(b^2 - 4ac)) / 2a, cancels when the ray runs nearly
        // along the mantle (a small).  Rays with |a| <= 1e-8 are traced without bounds (gmlrt_kernels.hip:
        // cone_special_case); above that the root is off by at most 2^-52 * 2|b| / 2e-8 = 2.2e-8 |b|, |b| <= 2 |origin|
        // in object space: a thousandth of the unit cone, as for the cylinder, covers origins 2e4 object units away.
        lo[0] = lo[2] = -1.001; hi[0] = hi[2] = 1.001; lo[1] = -0.001; hi[1] = 1.001;
        break;
      default: return plane_bound(t);  // half-space: at best bounded on one side of one axis
    }
    const double (*M)[4] = t.m.m;
    const double lc[3] = {0.5 * (lo[0] + hi[0]), 0.5 * (lo[1] + hi[1]), 0.5 * (lo[2] + hi[2])};
    for (int i = 0; i < 3; i++) b.c[i] = M[i][0] * lc[0] + M[i][1] * lc[1] + M[i][2] * lc[2] + M[i][3];
    // An affine map takes the box to a parallelepiped; the farthest of its corners from the
    // image of the centre bounds everything inside it.
    double r2 = 0.0;
    for (int corner = 0; corner < 8; corner++) {
      double e[3];
      for (int k = 0; k < 3; k++) e[k] = ((corner >> k) & 1 ? hi[k] : lo[k]) - lc[k];
      double q = 0.0;
      for (int i = 0; i < 3; i++) {
        const double w = M[i][0] * e[0] + M[i][1] * e[1] + M[i][2] * e[2];
        q += w * w;
      }
      r2 = std::max(r2, q);
    }
    double r = std::sqrt(r2);
    if (prim == gml::PrimType::Sphere) {
      // the unit ball itself: its image reaches no farther than the largest stretch of the linear
      // part, lambda_max(M^T M) <= the largest absolute row sum of M^T M
      double s2 = 0.0;
      for (int i = 0; i < 3; i++) {
        double row = 0.0;
        for (int j = 0; j < 3; j++) {
          double e = 0.0;
          for (int k = 0; k < 3; k++) e += M[k][i] * M[k][j];
          row += std::fabs(e);
        }
        s2 = std::max(s2, row);
      }
      r = std::min(r, std::sqrt(s2));
    }
    b.r = r * (1.0 + 1e-6) + 1e-9;
    b.finite = std::isfinite(b.r) && std::isfinite(b.c[0]) && std::isfinite(b.c[1]) && std::isfinite(b.c[2]);
    // the box: extent of the parallelepiped along each world axis (for the ball: of the ellipsoid)
    for (int i = 0; i < 3; i++) {
      double ext;
      if (prim == gml::PrimType::Sphere) {
        ext = std::sqrt(M[i][0] * M[i][0] + M[i][1] * M[i][1] + M[i][2] * M[i][2]);
      } else {
        ext = 0.0;
        for (int k = 0; k < 3; k++) ext += std::fabs(M[i][k]) * 0.5 * (hi[k] - lo[k]);
      }
      const double slack = ext * 1e-6 + std::fabs(b.c[i]) * 1e-9 + 1e-9;
      b.lo[i] = b.c[i] - ext - slack;
      b.hi[i] = b.c[i] + ext + slack;
      if (!std::isfinite(b.lo[i]) || !std::isfinite(b.hi[i])) { b.lo[i] = -kInf; b.hi[i] = kInf; }
    }
    if (b.finite) tighten(b);
    return b;
  }
  // The solid of a plane is y <= 0 in object space (primitives.rs:924-931), i.e.
  // n.p + t <= 0 in world space with (n, t) = row 1 of the inverse matrix.  When n is
  // parallel to a world axis that is a bound on one side of that axis.
  static Bound plane_bound(const gml::Transform& t) {
    Bound b;
    const double* row = t.inv_m.m[1];
    int axis = -1;
    for (int i = 0; i < 3; i++)
      if (row[i] != 0.0) { if (axis >= 0) return b; axis = i; }
    if (axis < 0 || !std::isfinite(row[axis]) || !std::isfinite(row[3])) return b;
    const double edge = -row[3] / row[axis];
    const double slack = std::fabs(edge) * 1e-6 + 1e-9;
    if (!std::isfinite(edge)) return b;
    if (row[axis] > 0.0) b.hi[axis] = edge + slack;
    else b.lo[axis] = edge - slack;
    return b;
  }
  static Bound sphere_union(const Bound& p, const Bound& q) {
    Bound u;
    if (!p.finite || !q.finite) return u;
    const double d[3] = {q.c[0] - p.c[0], q.c[1] - p.c[1], q.c[2] - p.c[2]};
    const double dist = std::sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
    if (dist + q.r <= p.r) return p;
    if (dist + p.r <= q.r) return q;
    u.r = 0.5 * (dist + p.r + q.r);
    const double f = dist > 0.0 ? (u.r - p.r) / dist : 0.0;
    for (int i = 0; i < 3; i++) u.c[i] = p.c[i] + d[i] * f;
    u.r = u.r * (1.0 + 1e-9) + 1e-12;
    u.finite = std::isfinite(u.r);
    return u;
  }

  // Emits the postfix program of subtree `n`.  A bound word goes in front of it when
  // its volume is finite and clearly tighter than the nearest enclosing emitted one
  // (testing nested volumes of almost the same size costs more than it saves).
  static double ratio_from_env(const char* name, double dflt) {
    const char* e = std::getenv(name);
    if (!e || !*e) return dflt;
    const double v = std::atof(e);
    return v > 0.0 ? v : dflt;
  }
  // surface-area ratios; the environment overrides exist for tuning runs
  static double op_ratio() { static const double v = ratio_from_env("GMLRT_BOUND_OP_RATIO", 0.8); return v; }
  static double leaf_ratio() { static const double v = ratio_from_env("GMLRT_BOUND_LEAF_RATIO", 0.8); return v; }
  // `frontier`: the node hangs from the root through unions only.  The first non-union node
  // on such a path (an operand of the root's union tree) always gets its bound when it has a
  // finite one: a consumer that regroups those operands spatially needs each one's volume.
  void emit(int ni, double enclosing_area, bool frontier) {
    const FNode& n = tree[(size_t)ni];
    const bool is_leaf = n.leaf >= 0;
    const bool is_union = !is_leaf && n.rule == gml::Rule::Union;
    const uint64_t worst_words = 2ull * (2ull * n.n_leaves - 1ull);
    const double area = n.bnd.area();
    bool want = n.bnd.finite && bounds.size() < GMLRT_BOUND_MAX && worst_words <= GMLRT_BOUND_MAX;
    if (want && std::isfinite(enclosing_area) && !(frontier && !is_union))
      want = area <= (is_leaf ? leaf_ratio() : op_ratio()) * enclosing_area;
    size_t at = 0;
    if (want) {
      at = nodes.size();
      nodes.push_back(0);
      gmlrt_bound g;
      for (int i = 0; i < 3; i++) { g.center[i] = n.bnd.c[i]; g.lo[i] = n.bnd.lo[i]; g.hi[i] = n.bnd.hi[i]; }
      g.radius = n.bnd.r;
      bounds.push_back(g);
    }
    const size_t my_bound = bounds.size() - 1;
    const double inner_r = want ? area : enclosing_area;
    if (is_leaf) {
      nodes.push_back(GMLRT_NODE_LEAF(n.leaf));
      ostack.emplace_back(true, 2);
      sp += 2;
      max_events = std::max(max_events, sp);
      max_stack = std::max(max_stack, ostack.size());
    } else {
      emit(n.a, inner_r, frontier && is_union);
      emit(n.b, inner_r, frontier && is_union);
      auto b = ostack.back(); ostack.pop_back();
      auto a = ostack.back(); ostack.pop_back();
      nodes.push_back(GMLRT_NODE_OP((uint32_t)n.rule, a.first, b.first));
      // the merge writes its output above both operands before moving it down
      max_events = std::max(max_events, sp + a.second + b.second);
      ostack.emplace_back(false, a.second + b.second);
    }
    if (want) nodes[at] = GMLRT_NODE_BOUND(my_bound, nodes.size() - at - 1);
  }
};

template <class T>
void append(std::vector<uint8_t>& blob, uint64_t* off, const std::vector<T>& v) {
  while (blob.size() % 16) blob.push_back(0);
  *off = blob.size();
  const uint8_t* p = (const uint8_t*)v.data();
  blob.insert(blob.end(), p, p + v.size() * sizeof(T));
}

}  // namespace

std::vector<uint8_t> flatten_scene(const gml::RenderCall& rc) {
  Flattener f;
  gml::walk_postfix(rc.scene, f);
  f.emit(f.tstack.back(), std::numeric_limits<double>::infinity(), true);
  std::vector<gmlrt_light> lights;
  for (auto& l : rc.lights) {
    gmlrt_light g;
    std::memset(&g, 0, sizeof(g));
    g.kind = (uint32_t)l->kind;
    for (int i = 0; i < 3; i++) { g.dir[i] = l->dir[(size_t)i]; g.pos[i] = l->pos[(size_t)i]; g.color[i] = l->color[(size_t)i]; }
    g.coscutoff = l->coscutoff;
    g.exp = l->exp;
    lights.push_back(g);
  }
  gmlrt_blob_header h;
  std::memset(&h, 0, sizeof(h));
  h.magic = GMLRT_BLOB_MAGIC;
  h.version = GMLRT_BLOB_VERSION;
  h.n_leaves = (uint32_t)f.leaves.size();
  h.n_nodes = (uint32_t)f.nodes.size();
  h.n_lights = (uint32_t)lights.size();
  h.n_code_words = (uint32_t)f.code.size();
  h.n_consts = (uint32_t)f.consts.size();
  h.max_stack = (uint32_t)f.max_stack;
  if (f.max_events > 0xFFFFFFFFull) throw SceneTooLarge("scene too large");
  if (2ull * f.leaves.size() > kMaxListEvents) throw SceneTooLarge("a CSG node can yield more than 32767 events per ray");
  if (lights.size() > kMaxLights) throw SceneTooLarge("more than 2047 lights");
  h.max_events = (uint32_t)f.max_events;
  h.n_bounds = (uint32_t)f.bounds.size();
  std::vector<uint8_t> blob(sizeof(h), 0);
  append(blob, &h.off_leaves, f.leaves);
  append(blob, &h.off_nodes, f.nodes);
  append(blob, &h.off_lights, lights);
  append(blob, &h.off_code, f.code);
  append(blob, &h.off_consts, f.consts);
  append(blob, &h.off_bounds, f.bounds);
  while (blob.size() % 16) blob.push_back(0);
  h.total_bytes = blob.size();
  std::memcpy(blob.data(), &h, sizeof(h));
  return blob;
}

// ============================================================== validation
std::string validate_blob(const void* blob, size_t bytes) {
  if (!blob || bytes < sizeof(gmlrt_blob_header)) return "blob shorter than its header";
  gmlrt_blob_header h;
  std::memcpy(&h, blob, sizeof(h));
  if (h.magic != GMLRT_BLOB_MAGIC) return "bad magic";
  if (h.version != GMLRT_BLOB_VERSION) return "unsupported blob version";
  if (h.total_bytes != bytes) return "total_bytes does not match the buffer size";
  if (h.n_leaves == 0 || h.n_leaves > (1u << 20)) return "leaf count out of range";
  // Widths of the kernel's packed per-lane fields (the reference has neither limit): the light a
  // shadow ray belongs to travels in 11 bits of the lane state, and the length of a list on the
  // operand stack in 15 bits (a list holds at most two events per leaf below its node).
  if (h.n_lights > kMaxLights) return "too large: more than 2047 lights";
  if (h.n_bounds > GMLRT_BOUND_MAX + 1u) return "too many bounds";
  if (h.n_nodes != 2 * h.n_leaves - 1 + h.n_bounds) return "node count is not 2*leaves-1+bounds";
  auto section_ok = [&](uint64_t off, uint64_t count, uint64_t elem) {
    return off % 8 == 0 && off >= sizeof(h) && off <= bytes && count * elem <= bytes - off;
  };
  if (!section_ok(h.off_leaves, h.n_leaves, sizeof(gmlrt_leaf))) return "leaf section out of bounds";
  if (!section_ok(h.off_nodes, h.n_nodes, 4)) return "node section out of bounds";
  if (!section_ok(h.off_lights, h.n_lights, sizeof(gmlrt_light))) return "light section out of bounds";
  if (!section_ok(h.off_code, h.n_code_words, 8)) return "code section out of bounds";
  if (!section_ok(h.off_consts, h.n_consts, 8)) return "constant section out of bounds";
  if (!section_ok(h.off_bounds, h.n_bounds, sizeof(gmlrt_bound))) return "bounds section out of bounds";
  const uint8_t* base = (const uint8_t*)blob;
  const gmlrt_bound* bnds = (const gmlrt_bound*)(base + h.off_bounds);
  for (uint32_t i = 0; i < h.n_bounds; i++)
    if (!(bnds[i].radius >= 0.0) || !std::isfinite(bnds[i].radius) || !std::isfinite(bnds[i].center[0]) ||
        !std::isfinite(bnds[i].center[1]) || !std::isfinite(bnds[i].center[2]))
      return "bound is not a finite sphere";
  for (uint32_t i = 0; i < h.n_bounds; i++)
    for (int k = 0; k < 3; k++)
      if (bnds[i].lo[k] != bnds[i].lo[k] || bnds[i].hi[k] != bnds[i].hi[k]) return "bound box side is NaN";
  // open bound words: (index of the first word after the subtree, operand-stack depth when it opened)
  std::vector<std::pair<uint32_t, size_t>> open_bounds;
  uint32_t bounds_seen = 0;
  const gmlrt_leaf* leaves = (const gmlrt_leaf*)(base + h.off_leaves);
  const uint32_t* nodes = (const uint32_t*)(base + h.off_nodes);
  const uint64_t* code = (const uint64_t*)(base + h.off_code);
  const gmlrt_light* lights = (const gmlrt_light*)(base + h.off_lights);
  // postfix program: every leaf used exactly once and in order, stack discipline,
  // operand-kind flags and the two capacity fields must be what the traversal assumes
  std::vector<std::pair<bool, uint64_t>> st;
  uint64_t sp = 0, max_events = 0;
  size_t max_stack = 0;
  uint32_t next_leaf = 0;
  for (uint32_t i = 0; i < h.n_nodes; i++) {
    const uint32_t w = nodes[i];
    const uint32_t kind = w >> 30;
    if (kind == 0) {
      if ((w & 0x3FFFFFFFu) != next_leaf) return "leaves must appear in index order";
      next_leaf++;
      st.emplace_back(true, 2);
      sp += 2;
      max_events = std::max(max_events, sp);
      max_stack = std::max(max_stack, st.size());
    } else if (kind == 1) {
      if ((w & 3u) > 2u || (w & 0x3FFFFFF0u) != 0) return "bad operator word";
      if (st.size() < 2) return "postfix program underflows its stack";
      if (!open_bounds.empty() && st.size() < open_bounds.back().second + 2)
        return "bound does not cover exactly one subtree";
      auto b = st.back(); st.pop_back();
      auto a = st.back(); st.pop_back();
      if (((w >> 2) & 1u) != (a.first ? 1u : 0u) || ((w >> 3) & 1u) != (b.first ? 1u : 0u))
        return "operator leaf-operand flags do not match the tree";
      max_events = std::max(max_events, sp + a.second + b.second);
      if (a.second + b.second > kMaxListEvents) return "too large: a CSG node can yield more than 32767 events per ray";
      st.emplace_back(false, a.second + b.second);
    } else if (kind == 2) {
      // the words a bound covers must be exactly one complete subtree: the kernel
      // replaces them by one empty list and continues right after them
      const uint32_t idx = (w >> 15) & 0x7FFFu, nwords = w & 0x7FFFu;
      if (idx >= h.n_bounds) return "bound index out of range";
      if (nwords == 0 || (uint64_t)i + 1 + nwords > h.n_nodes) return "bound covers words past the program";
      if (!open_bounds.empty() && i + 1 + nwords > open_bounds.back().first) return "bound crosses an enclosing bound";
      open_bounds.emplace_back(i + 1 + nwords, st.size());
      bounds_seen++;
      continue;
    } else {
      return "bad node word";
    }
    while (!open_bounds.empty() && open_bounds.back().first == i + 1) {
      if (st.size() != open_bounds.back().second + 1) return "bound does not cover exactly one subtree";
      open_bounds.pop_back();
    }
  }
  if (!open_bounds.empty() || bounds_seen != h.n_bounds) return "bound words do not match the bounds section";
  if (st.size() != 1 || next_leaf != h.n_leaves) return "postfix program does not reduce to one root";
  if (h.max_stack < max_stack) return "max_stack too small";
  if (h.max_events < max_events) return "max_events too small";
  for (uint32_t i = 0; i < h.n_lights; i++) {
    if (lights[i].kind > GMLRT_LIGHT_SPOT) return "bad light kind";
    // every ray the device traces has a unit direction (its bound tests rely on it); a directional
    // light's shadow rays take theirs from here (lights.rs:36 normalises it at construction)
    const double* dv = lights[i].dir;
    const double len2 = dv[0] * dv[0] + dv[1] * dv[1] + dv[2] * dv[2];
    if (lights[i].kind == GMLRT_LIGHT_DIRECTIONAL && std::fabs(len2 - 1.0) > 1e-9) return "directional light direction is not normalised";
  }
  for (uint32_t i = 0; i < h.n_leaves; i++) {
    const gmlrt_leaf& L = leaves[i];
    if (L.type > GMLRT_PLANE) return "bad primitive type";
    if (L.surf_code == GMLRT_SURF_CONST) {
      if ((uint64_t)L.surf_const + 6 > h.n_consts) return "constant surface out of bounds";
      continue;
    }
    if (L.surf_code >= h.n_code_words) return "surface code offset out of bounds";
    bool ended = false;
    for (uint32_t pc = L.surf_code; pc < h.n_code_words; pc++) {
      const uint64_t w = code[pc];
      const uint32_t op = (uint32_t)(w & 0xFF);
      if (op >= GMLRT_SOP__COUNT) return "bad surface opcode";
      if (op == GMLRT_SOP_END) { ended = true; break; }
      const uint32_t dst = (uint32_t)((w >> 8) & 0xFF);
      if (op == GMLRT_SOP_OUT ? (dst != 0 && dst != 3) : op == GMLRT_SOP_TRAP ? dst != 0 : dst >= GMLRT_SURF_MAX_REGS)
        return "bad surface destination";
      for (int k = 0; k < 3; k++) {
        const uint32_t o = (uint32_t)((w >> (16 + 16 * k)) & 0xFFFF);
        if (o & 0x8000u) {
          if ((uint64_t)L.surf_const + (o & 0x7FFFu) >= h.n_consts) return "surface constant out of bounds";
        } else if (o >= GMLRT_SURF_MAX_REGS) {
          return "surface register out of bounds";
        }
      }
    }
    if (!ended) return "surface code runs off the end of the section";
  }
  return "";
}

// The classes of ray directions that the kernel traces without bounds (gmlrt_kernels.hip: cylinder_special_case,
// cone_special_case; layout of a class there and at kAxisWords).  First the cylinders' axes and the normals of the
// cylinders' and cones' cap planes in world space, one class per distinct direction, then one quadratic form per distinct
// cone shape and orientation.
void special_ray_classes(const void* blob, std::vector<double>* classes, uint32_t* n_axes, uint32_t* n_cone_forms) {
  const gmlrt_blob_header& H = *(const gmlrt_blob_header*)blob;
  const gmlrt_leaf* L = (const gmlrt_leaf*)((const uint8_t*)blob + H.off_leaves);
  std::vector<double>& axes = *classes;
  axes.clear();
  for (uint32_t i = 0; i < H.n_leaves; i++) {
    // (a cone joins with the planes across its axis only: a ray that runs within its base plane, or within rounding
    // of it, gets the crossing of that plane, (1 - py) / dy, as the quotient of two rounding errors whenever the
    // height test of primitives.rs:820-827 keeps one of the two mantle crossings and not the other, which rounding
    // decides; the interval the reference then reports can reach far beyond the solid.  Found by the lattice fuzz: a
    // bounce ray in the plane x = 0, the base plane of a cone on its side.)
    if (L[i].type != GMLRT_CYLINDER && L[i].type != GMLRT_CONE) continue;
    // object direction u and world direction d of one ray: d = M u up to scale.  In the parallel case the part of u
    // across the axis is at most 4.473e-4 of |u| (1 - (1 - 1e-7)^2 = 2e-7), so d lies within asin(rho) of M e_y with
    // rho = |M|_2 * 4.473e-4 / (|M e_y| - |M|_2 * 4.473e-4); in the orthogonal case u's part along the axis is at most
    // 1e-7 of |u| = |A d| <= |A|_2 (A = the inverse's linear part), so |d . n| <= 1e-7 * |A|_2 / |A^T e_y| for the
    // unit normal n = A^T e_y / |A^T e_y| of the planes across the axis.  Axis and normal are classes of their own
    // (they differ for a sheared cylinder).  Margins of 2 %: the directions are compared in doubles, and every
    // flagged ray costs its wave a walk through the whole tree.
    double ax[3], nn[3];
    for (int r = 0; r < 3; r++) {
      ax[r] = L[i].m[r * 4 + 1];
      nn[r] = L[i].inv_m[4 + r];
    }
    // (the largest stretch of a 3 x 4 transform's linear part: sqrt(lambda_max(T^T T)) <= the square root of the
    // largest absolute row sum of T^T T, with equality for rotations, scalings along the axes and their products)
    auto stretch = [](const double* T) {
      double top = 0.0;
      for (int a = 0; a < 3; a++) {
        double row = 0.0;
        for (int b = 0; b < 3; b++) row += std::fabs(T[a] * T[b] + T[4 + a] * T[4 + b] + T[8 + a] * T[8 + b]);
        top = std::max(top, row);
      }
      return std::sqrt(top);
    };
    const double fm = stretch(L[i].m), fa = stretch(L[i].inv_m);
    const double la = std::sqrt(ax[0] * ax[0] + ax[1] * ax[1] + ax[2] * ax[2]), ln = std::sqrt(nn[0] * nn[0] + nn[1] * nn[1] + nn[2] * nn[2]);
    double hi = -1.0, lo = 2.0;  // (a degenerate transform: every ray takes the exact path)
    if (la > 0.0 && ln > 0.0 && std::isfinite(fm) && std::isfinite(fa)) {
      const double spread = fm * 4.48e-4 * 1.02;
      const double rho = la > spread ? spread / (la - spread) : 1.0;
      hi = rho < 1.0 ? std::sqrt(1.0 - rho * rho) : -1.0;
      lo = 1e-7 * fa * 1.02 / ln;  // bound on |d . n| for n of unit length: |(A d)_y| < 1e-7 |A d| <= 1e-7 |A|_2
      if (!(lo < 1.0)) lo = 2.0;
      for (int r = 0; r < 3; r++) { ax[r] /= la; nn[r] /= ln; }
    }
    auto add = [&](const double v[3], double h_, double l_) {
      for (size_t k = 0; k < axes.size(); k += kAxisWords) {
        const double dot = std::fabs(axes[k] * v[0] + axes[k + 1] * v[1] + axes[k + 2] * v[2]);
        if (dot >= 1.0 - 1e-12) {  // the same direction: one class, the wider margins
          axes[k + 3] = std::min(axes[k + 3], h_);
          axes[k + 4] = std::max(axes[k + 4], l_);
          return;
        }
      }
      axes.insert(axes.end(), {v[0], v[1], v[2], h_, l_, 0.0, 0.0, 0.0});
    };
    if (L[i].type == GMLRT_CYLINDER) add(ax, hi, -1.0);   // parallel to the axis
    add(nn, 2.0, lo);    // across it: orthogonal to the normal of the cap planes
  }
  *n_axes = (uint32_t)(axes.size() / kAxisWords);
  // The cones' forms (see cone_special_case), one per distinct shape and orientation.
  *n_cone_forms = 0;
  for (uint32_t i = 0; i < H.n_leaves; i++) {
    if (L[i].type != GMLRT_CONE) continue;
    double M[3][3], Q[3][3], tr = 0.0;
    for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) M[r][c] = L[i].inv_m[r * 4 + c];
    for (int a = 0; a < 3; a++)
      for (int b = 0; b < 3; b++) {
        Q[a][b] = M[0][a] * M[0][b] - M[1][a] * M[1][b] + M[2][a] * M[2][b];  // (M^T J M)
        if (a == b) tr += M[0][a] * M[0][a] + M[1][a] * M[1][a] + M[2][a] * M[2][a];
      }
    double f[kAxisWords] = {Q[0][0] / tr, Q[1][1] / tr, Q[2][2] / tr, 2.0 * Q[0][1] / tr, 2.0 * Q[0][2] / tr, 2.0 * Q[1][2] / tr, kConeFormBound, 0.0};
    bool finite = tr > 0.0;
    for (int k = 0; k < 6; k++) finite = finite && std::isfinite(f[k]);
    if (!finite) { for (int k = 0; k < 6; k++) f[k] = 0.0; }  // (a degenerate transform: every ray takes the exact path)
    bool have = false;
    for (size_t k = (size_t)*n_axes * kAxisWords; k < axes.size() && !have; k += kAxisWords) {
      have = true;
      for (int q = 0; q < 6; q++) have = have && std::fabs(axes[k + (size_t)q] - f[q]) <= 1e-12;
    }
    if (!have) { axes.insert(axes.end(), f, f + kAxisWords); (*n_cone_forms)++; }
  }
}

}  // namespace gmlrt
