#include "gml_front.hpp"

#include <pthread.h>

#include <cerrno>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <limits>

namespace gml {

// ============================================================== tokenizer
static const struct { const char* name; Op op; } kOps[] = {
    {"acos", Op::Acos}, {"addf", Op::Addf}, {"addi", Op::Addi}, {"apply", Op::Apply},
    {"asin", Op::Asin}, {"clampf", Op::Clampf}, {"cone", Op::Cone}, {"cos", Op::Cos},
    {"cube", Op::Cube}, {"cylinder", Op::Cylinder}, {"difference", Op::Difference},
    {"divf", Op::Divf}, {"divi", Op::Divi}, {"eqf", Op::Eqf}, {"eqi", Op::Eqi},
    {"floor", Op::Floor}, {"frac", Op::Frac}, {"get", Op::Get}, {"getx", Op::Getx},
    {"gety", Op::Gety}, {"getz", Op::Getz}, {"if", Op::If}, {"intersect", Op::Intersect},
    {"length", Op::Length}, {"lessf", Op::Lessf}, {"lessi", Op::Lessi}, {"light", Op::Light},
    {"modi", Op::Modi}, {"mulf", Op::Mulf}, {"muli", Op::Muli}, {"negf", Op::Negf},
    {"negi", Op::Negi}, {"plane", Op::Plane}, {"point", Op::Point},
    {"pointlight", Op::Pointlight}, {"real", Op::Real}, {"render", Op::Render},
    {"rotatex", Op::Rotatex}, {"rotatey", Op::Rotatey}, {"rotatez", Op::Rotatez},
    {"scale", Op::Scale}, {"sin", Op::Sin}, {"sphere", Op::Sphere},
    {"spotlight", Op::Spotlight}, {"sqrt", Op::Sqrt}, {"subf", Op::Subf}, {"subi", Op::Subi},
    {"translate", Op::Translate}, {"union", Op::Union}, {"uscale", Op::Uscale},
};

const char* op_name(Op op) {
  for (auto& e : kOps)
    if (e.op == op) return e.name;
  return "?";
}

bool match_operator(const std::string& s, Op* out) {
  for (auto& e : kOps)
    if (s == e.name) {
      if (out) *out = e.op;
      return true;
    }
  return false;
}

bool Token::operator==(const Token& o) const {
  if (kind != o.kind) return false;
  switch (kind) {
    case Identifier: case Binder: case Str: return s == o.s;
    case Operator: return op == o.op;
    case Boolean: return b == o.b;
    case Real: return r == o.r;
    case Integer: return i == o.i;
    default: return true;
  }
}

namespace {

bool is_ws(char c) { return c == ' ' || (c >= '\t' && c <= '\r'); }
bool is_id_start(char c) { return (c >= 'a' && c <= 'z') || (c >= 'A' && c <= 'Z'); }
bool is_id_rest(char c) { return is_id_start(c) || (c >= '0' && c <= '9') || c == '-' || c == '_'; }
bool is_digit(char c) { return c >= '0' && c <= '9'; }
bool is_exp(char c) { return c == 'e' || c == 'E'; }

size_t eat_digits(const char* a, size_t n) {
  size_t k = 0;
  while (k < n && is_digit(a[k])) k++;
  return k;
}

// length of the identifier at a[0..n), 0 if none (src/tokenizer.rs:262-279)
size_t match_identifier(const char* a, size_t n) {
  if (n == 0 || !is_id_start(a[0])) return 0;
  size_t k = 0;
  while (k < n && is_id_rest(a[k])) k++;
  return k;
}

// src/tokenizer.rs:326-387; returns consumed length or 0
size_t match_real(const char* a, size_t n) {
  size_t c = 0;
  if (n == 0) return 0;
  if (a[0] == '-') {
    c = 1;
    if (n == c) return 0;
  }
  size_t d = eat_digits(a + c, n - c);
  if (d == 0) return 0;
  c += d;
  if (n == c) return 0;
  if (a[c] == '.') {
    c += 1;
    if (n == c) return 0;
    size_t dec = eat_digits(a + c, n - c);
    if (dec == 0) return 0;
    c += dec;
  } else if (!is_exp(a[c])) {
    return 0;
  }
  if (c < n && is_exp(a[c])) {
    c += 1;
    if (n == c) return 0;
    if (a[c] == '-') {
      c += 1;
      if (n == c) return 0;
    }
    size_t e = eat_digits(a + c, n - c);
    if (e == 0) return 0;
    c += e;
  }
  return c;
}

}  // namespace

std::vector<Token> tokenize(const std::string& text) {
  std::vector<Token> out;
  const char* t = text.data();
  const size_t len = text.size();
  size_t pos = 0;
  while (true) {
    const size_t last = pos;
    const char* a = t + pos;
    const size_t n = len - pos;
    do {
      if (n == 0) break;
      // whitespace
      if (is_ws(a[0])) {
        size_t k = 0;
        while (k < n && is_ws(a[k])) k++;
        pos += k;
        break;
      }
      // block comment; an unterminated one aborts the reference, here it ends the scan
      if (n >= 2 && a[0] == '/' && a[1] == '*') {
        size_t k = 2;
        while (k + 1 < n && !(a[k] == '*' && a[k + 1] == '/')) k++;
        if (k + 1 >= n) throw GmlError("unterminated block comment");
        pos += k + 2;
        break;
      }
      // line comment
      if (a[0] == '%') {
        size_t k = 0;
        while (k < n && a[k] != '\n') k++;
        pos += k;
        break;
      }
      Token tok;
      if (a[0] == '{') { tok.kind = Token::BeginFunction; out.push_back(tok); pos += 1; break; }
      if (a[0] == '}') { tok.kind = Token::EndFunction; out.push_back(tok); pos += 1; break; }
      if (a[0] == '[') { tok.kind = Token::BeginArray; out.push_back(tok); pos += 1; break; }
      if (a[0] == ']') { tok.kind = Token::EndArray; out.push_back(tok); pos += 1; break; }
      // booleans fire on the bare prefix (src/tokenizer.rs:155-163)
      if (n >= 4 && std::memcmp(a, "true", 4) == 0) {
        tok.kind = Token::Boolean; tok.b = true; out.push_back(tok); pos += 4; break;
      }
      if (n >= 5 && std::memcmp(a, "false", 5) == 0) {
        tok.kind = Token::Boolean; tok.b = false; out.push_back(tok); pos += 5; break;
      }
      if (size_t k = match_identifier(a, n)) {
        std::string id(a, k);
        Op op;
        if (match_operator(id, &op)) { tok.kind = Token::Operator; tok.op = op; }
        else { tok.kind = Token::Identifier; tok.s = id; }
        out.push_back(tok);
        pos += k;
        break;
      }
      if (a[0] == '/' && n > 1) {
        size_t k = match_identifier(a + 1, n - 1);
        if (k) {
          std::string id(a + 1, k);
          if (!match_operator(id, nullptr)) {
            tok.kind = Token::Binder; tok.s = id; out.push_back(tok); pos += k + 1; break;
          }
        }
      }
      if (size_t k = match_real(a, n)) {
        tok.kind = Token::Real;
        tok.r = std::strtod(std::string(a, k).c_str(), nullptr);
        out.push_back(tok);
        pos += k;
        break;
      }
      {
        size_t p = (a[0] == '-') ? 1 : 0;
        size_t d = eat_digits(a + p, n - p);
        if (d) {
          std::string s(a, p + d);
          errno = 0;
          long long v = std::strtoll(s.c_str(), nullptr, 10);
          if (errno == ERANGE) throw GmlError("error parsing integer");
          tok.kind = Token::Integer; tok.i = v; out.push_back(tok); pos += p + d; break;
        }
      }
      if (n > 1 && a[0] == '"') {
        size_t k = 0;
        while (1 + k < n && a[1 + k] != '"') k++;
        if (k > 0) {
          if (1 + k >= n) throw GmlError("unterminated string");
          tok.kind = Token::Str; tok.s.assign(a + 1, k); out.push_back(tok); pos += k + 2; break;
        }
      }
    } while (false);
    if (pos == last || pos == len) break;
  }
  return out;
}

// ============================================================== parser
namespace {
struct ParseOut { size_t i; std::shared_ptr<AstList> ast; };

ParseOut do_parse(const std::vector<Token>& tokens, size_t offset) {
  auto ast = std::make_shared<AstList>();
  size_t i = offset;
  while (i < tokens.size()) {
    const Token& t = tokens[i];
    if (t.kind == Token::EndFunction || t.kind == Token::EndArray) return {i + 1, ast};
    if (t.kind == Token::BeginFunction || t.kind == Token::BeginArray) {
      ParseOut sub = do_parse(tokens, i + 1);
      i = sub.i;
      const Token::Kind want = t.kind == Token::BeginFunction ? Token::EndFunction : Token::EndArray;
      if (tokens[i - 1].kind != want) throw GmlError("SyntaxError");
      AstNode n;
      n.kind = t.kind == Token::BeginFunction ? AstNode::Function : AstNode::Array;
      n.children = sub.ast;
      ast->push_back(std::move(n));
    } else {
      AstNode n;
      n.kind = AstNode::Leaf;
      n.tok = t;
      ast->push_back(std::move(n));
      i += 1;
    }
  }
  return {i, ast};
}
}  // namespace

std::shared_ptr<const AstList> parse(const std::vector<Token>& tokens) {
  ParseOut r = do_parse(tokens, 0);
  if (r.i != tokens.size()) throw GmlError("SyntaxError");
  return r.ast;
}

// ============================================================== math / transform
Mat4 mat4_identity() {
  Mat4 r;
  for (int i = 0; i < 4; i++)
    for (int j = 0; j < 4; j++) r.m[i][j] = (i == j) ? 1.0 : 0.0;
  return r;
}

Mat4 mat4_mul(const Mat4& a, const Mat4& b) {
  Mat4 r;
  for (int i = 0; i < 4; i++)
    for (int j = 0; j < 4; j++)
      r.m[i][j] = a.m[i][0] * b.m[0][j] + a.m[i][1] * b.m[1][j] + a.m[i][2] * b.m[2][j] +
                  a.m[i][3] * b.m[3][j];
  return r;
}

static double to_radians(double d) { return d * (3.14159265358979323846 / 180.0); }
static double to_degrees(double r) { return r * (180.0 / 3.14159265358979323846); }

static Mat4 rows(std::initializer_list<double> v) {
  Mat4 r;
  auto it = v.begin();
  for (int i = 0; i < 4; i++)
    for (int j = 0; j < 4; j++) r.m[i][j] = *it++;
  return r;
}

void Transform::scale(double sx, double sy, double sz) {
  Mat4 sc = rows({sx, 0, 0, 0, 0, sy, 0, 0, 0, 0, sz, 0, 0, 0, 0, 1});
  Mat4 inv = rows({1.0 / sx, 0, 0, 0, 0, 1.0 / sy, 0, 0, 0, 0, 1.0 / sz, 0, 0, 0, 0, 1});
  m = mat4_mul(sc, m);
  inv_m = mat4_mul(inv_m, inv);
}
void Transform::translate(double tx, double ty, double tz) {
  Mat4 tr = rows({1, 0, 0, tx, 0, 1, 0, ty, 0, 0, 1, tz, 0, 0, 0, 1});
  Mat4 inv = rows({1, 0, 0, -tx, 0, 1, 0, -ty, 0, 0, 1, -tz, 0, 0, 0, 1});
  m = mat4_mul(tr, m);
  inv_m = mat4_mul(inv_m, inv);
}
void Transform::rotatex(double d) {
  double c = std::cos(to_radians(d)), s = std::sin(to_radians(d));
  Mat4 r = rows({1, 0, 0, 0, 0, c, -s, 0, 0, s, c, 0, 0, 0, 0, 1});
  Mat4 inv = rows({1, 0, 0, 0, 0, c, s, 0, 0, -s, c, 0, 0, 0, 0, 1});
  m = mat4_mul(r, m);
  inv_m = mat4_mul(inv_m, inv);
}
void Transform::rotatey(double d) {
  double c = std::cos(to_radians(d)), s = std::sin(to_radians(d));
  Mat4 r = rows({c, 0, s, 0, 0, 1, 0, 0, -s, 0, c, 0, 0, 0, 0, 1});
  Mat4 inv = rows({c, 0, -s, 0, 0, 1, 0, 0, s, 0, c, 0, 0, 0, 0, 1});
  m = mat4_mul(r, m);
  inv_m = mat4_mul(inv_m, inv);
}
void Transform::rotatez(double d) {
  double c = std::cos(to_radians(d)), s = std::sin(to_radians(d));
  Mat4 r = rows({c, -s, 0, 0, s, c, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1});
  Mat4 inv = rows({c, s, 0, 0, -s, c, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1});
  m = mat4_mul(r, m);
  inv_m = mat4_mul(inv_m, inv);
}

void apply_step(Transform& t, const XformStep& s) {
  switch (s.k) {
    case XformKind::Translate: t.translate(s.p[0], s.p[1], s.p[2]); break;
    case XformKind::Scale: t.scale(s.p[0], s.p[1], s.p[2]); break;
    case XformKind::Uscale: t.scale(s.p[0], s.p[0], s.p[0]); break;  // src/transform.rs:60-62
    case XformKind::Rotatex: t.rotatex(s.p[0]); break;
    case XformKind::Rotatey: t.rotatey(s.p[0]); break;
    case XformKind::Rotatez: t.rotatez(s.p[0]); break;
  }
}

namespace {
// Steps are collected outermost-first while descending; a leaf replays them
// innermost-first, which is the order the reference applied them to that leaf.
void walk_rec(const Node* n, std::vector<XformStep>& pending, TreeVisitor& v) {
  // Xform chains can be long (one wrapper per GML transform op); iterate through them.
  size_t pushed = 0;
  while (n->kind == Node::Xform) {
    XformStep s{n->xk, {n->p[0], n->p[1], n->p[2]}};
    pending.push_back(s);
    pushed++;
    n = n->child.get();
  }
  if (n->kind == Node::Leaf) {
    Transform t;
    for (size_t i = pending.size(); i-- > 0;) apply_step(t, pending[i]);
    v.leaf(n->prim, n->surface, t);
  } else {
    walk_rec(n->a.get(), pending, v);
    walk_rec(n->b.get(), pending, v);
    v.op(n->rule);
  }
  pending.resize(pending.size() - pushed);
}
}  // namespace

void walk_postfix(const NodePtr& root, TreeVisitor& v) {
  std::vector<XformStep> pending;
  walk_rec(root.get(), pending, v);
}

// ============================================================== env / values
const ValuePtr* Env::get(const std::string& k) const {
  auto it = current.find(k);
  if (it != current.end()) return &it->second;
  auto is = snap->find(k);
  if (is != snap->end()) return &is->second;
  return nullptr;
}

std::shared_ptr<const EnvMap> Env::snapshot() const {
  if (current.empty()) return snap;  // same contents as a merged copy
  auto m = std::make_shared<EnvMap>(current);
  for (auto& kv : *snap) (*m)[kv.first] = kv.second;  // old snapshot wins (src/evaluator.rs:92-95)
  return m;
}

ValuePtr make_int(int64_t v) { auto p = std::make_shared<Value>(); p->kind = Value::Integer; p->i = v; return p; }
ValuePtr make_real(double v) { auto p = std::make_shared<Value>(); p->kind = Value::Real; p->r = v; return p; }
ValuePtr make_bool(bool v) { auto p = std::make_shared<Value>(); p->kind = Value::Boolean; p->b = v; return p; }
ValuePtr make_point(const Vec3& v) { auto p = std::make_shared<Value>(); p->kind = Value::Point; p->p = v; return p; }

bool value_equal(const Value& a, const Value& b) {
  if (a.kind != b.kind) return false;
  switch (a.kind) {
    case Value::Boolean: return a.b == b.b;
    case Value::Real: return a.r == b.r;
    case Value::Integer: return a.i == b.i;
    case Value::String: return a.s == b.s;
    case Value::Point: return a.p == b.p;
    case Value::Array:
      if (a.arr.size() != b.arr.size()) return false;
      for (size_t i = 0; i < a.arr.size(); i++)
        if (!value_equal(*a.arr[i], *b.arr[i])) return false;
      return true;
    default: return false;
  }
}

// ============================================================== scalar ops
int64_t divi(int64_t a, int64_t b) {
  if (b == 0) throw GmlError("attempt to divide by zero");
  if (a == std::numeric_limits<int64_t>::min() && b == -1) throw GmlError("attempt to divide with overflow");
  return a / b;
}
int64_t modi(int64_t a, int64_t b) { return a - b * divi(a, b); }
double gml_acos(double x) { return to_degrees(std::acos(x)); }
double gml_asin(double x) { return to_degrees(std::asin(x)); }
double gml_clampf(double x) { return std::fmin(std::fmax(x, 0.0), 1.0); }
double gml_cos(double x) { double r = std::cos(to_radians(x)); return std::fabs(r) < 1e-15 ? 0.0 : r; }
double gml_sin(double x) { double r = std::sin(to_radians(x)); return std::fabs(r) < 1e-15 ? 0.0 : r; }
int64_t gml_floor(double x) {
  double f = std::floor(x);
  if (f != f) return 0;  // saturating `as i64`
  if (f >= 9223372036854775808.0) return std::numeric_limits<int64_t>::max();
  if (f <= -9223372036854775808.0) return std::numeric_limits<int64_t>::min();
  return (int64_t)f;
}
double gml_frac(double x) { return x - std::trunc(x); }

// ============================================================== evaluator
namespace {
ValuePtr pop(Stack& s) {
  if (s.empty()) throw GmlError("EmptyStack");
  ValuePtr v = std::move(s.back());
  s.pop_back();
  return v;
}
int64_t get_integer(const ValuePtr& v) { if (v->kind != Value::Integer) throw GmlError("WrongTypeRef"); return v->i; }
double get_real(const ValuePtr& v) { if (v->kind != Value::Real) throw GmlError("WrongTypeRef"); return v->r; }
bool get_boolean(const ValuePtr& v) { if (v->kind != Value::Boolean) throw GmlError("WrongTypeRef"); return v->b; }
const Vec3& get_point(const ValuePtr& v) { if (v->kind != Value::Point) throw GmlError("WrongTypeRef"); return v->p; }
const NodePtr& get_node(const ValuePtr& v) { if (v->kind != Value::NodeV) throw GmlError("WrongTypeRef"); return v->node; }

ValuePtr make_node(NodePtr n) {
  auto p = std::make_shared<Value>();
  p->kind = Value::NodeV;
  p->node = std::move(n);
  return p;
}

Vec3 normalize(const Vec3& a) {
  double s = 1.0 / std::sqrt(a[0] * a[0] + a[1] * a[1] + a[2] * a[2]);
  return {a[0] * s, a[1] * s, a[2] * s};
}

void push_prim(Stack& stack, PrimType t) {
  ValuePtr surface = pop(stack);
  if (surface->kind != Value::Closure) throw GmlError("WrongTypeRef");
  auto n = std::make_shared<Node>();
  n->kind = Node::Leaf;
  n->prim = t;
  n->surface = surface;
  stack.push_back(make_node(n));
}

void push_op(Stack& stack, Rule r) {
  ValuePtr o2 = pop(stack), o1 = pop(stack);
  auto n = std::make_shared<Node>();
  n->kind = Node::Operator;
  n->rule = r;
  n->a = get_node(o1);
  n->b = get_node(o2);
  stack.push_back(make_node(n));
}

void push_xform(Stack& stack, XformKind k, int nargs) {
  double p[3] = {0, 0, 0};
  ValuePtr args[3];
  for (int i = nargs - 1; i >= 0; i--) args[i] = pop(stack);
  ValuePtr obj = pop(stack);
  if (obj->kind != Value::NodeV) throw GmlError("WrongType");
  for (int i = 0; i < nargs; i++) p[i] = get_real(args[i]);
  auto n = std::make_shared<Node>();
  n->kind = Node::Xform;
  n->xk = k;
  n->p[0] = p[0]; n->p[1] = p[1]; n->p[2] = p[2];
  n->child = obj->node;
  stack.push_back(make_node(n));
}
}  // namespace

void Evaluator::eval_op(Op op, Stack& stack) {
  switch (op) {
    case Op::Addi: { auto b = pop(stack), a = pop(stack); stack.push_back(make_int((int64_t)((uint64_t)get_integer(a) + (uint64_t)get_integer(b)))); break; }
    case Op::Subi: { auto b = pop(stack), a = pop(stack); stack.push_back(make_int((int64_t)((uint64_t)get_integer(a) - (uint64_t)get_integer(b)))); break; }
    case Op::Muli: { auto b = pop(stack), a = pop(stack); stack.push_back(make_int((int64_t)((uint64_t)get_integer(a) * (uint64_t)get_integer(b)))); break; }
    case Op::Divi: { auto b = pop(stack), a = pop(stack); stack.push_back(make_int(divi(get_integer(a), get_integer(b)))); break; }
    case Op::Modi: { auto b = pop(stack), a = pop(stack); stack.push_back(make_int(modi(get_integer(a), get_integer(b)))); break; }
    case Op::Eqi: { auto b = pop(stack), a = pop(stack); stack.push_back(make_bool(get_integer(a) == get_integer(b))); break; }
    case Op::Lessi: { auto b = pop(stack), a = pop(stack); stack.push_back(make_bool(get_integer(a) < get_integer(b))); break; }
    case Op::Negi: { auto a = pop(stack); stack.push_back(make_int(get_integer(a))); break; }  // no-op, src/evaluator.rs:446-450
    case Op::Real: { auto a = pop(stack); stack.push_back(make_real((double)get_integer(a))); break; }
    case Op::Addf: { auto b = pop(stack), a = pop(stack); stack.push_back(make_real(get_real(a) + get_real(b))); break; }
    case Op::Subf: { auto b = pop(stack), a = pop(stack); stack.push_back(make_real(get_real(a) - get_real(b))); break; }
    case Op::Mulf: { auto b = pop(stack), a = pop(stack); stack.push_back(make_real(get_real(a) * get_real(b))); break; }
    case Op::Divf: { auto b = pop(stack), a = pop(stack); stack.push_back(make_real(get_real(a) / get_real(b))); break; }
    case Op::Eqf: { auto b = pop(stack), a = pop(stack); stack.push_back(make_bool(get_real(a) == get_real(b))); break; }
    case Op::Lessf: { auto b = pop(stack), a = pop(stack); stack.push_back(make_bool(get_real(a) < get_real(b))); break; }
    case Op::Negf: { auto a = pop(stack); stack.push_back(make_real(get_real(a))); break; }  // no-op, src/evaluator.rs:452-456
    case Op::Acos: { auto a = pop(stack); stack.push_back(make_real(gml_acos(get_real(a)))); break; }
    case Op::Asin: { auto a = pop(stack); stack.push_back(make_real(gml_asin(get_real(a)))); break; }
    case Op::Clampf: { auto a = pop(stack); stack.push_back(make_real(gml_clampf(get_real(a)))); break; }
    case Op::Cos: { auto a = pop(stack); stack.push_back(make_real(gml_cos(get_real(a)))); break; }
    case Op::Sin: { auto a = pop(stack); stack.push_back(make_real(gml_sin(get_real(a)))); break; }
    case Op::Sqrt: { auto a = pop(stack); stack.push_back(make_real(std::sqrt(get_real(a)))); break; }
    case Op::Floor: { auto a = pop(stack); stack.push_back(make_int(gml_floor(get_real(a)))); break; }
    case Op::Frac: { auto a = pop(stack); stack.push_back(make_real(gml_frac(get_real(a)))); break; }
    case Op::Point: {
      auto z = pop(stack), y = pop(stack), x = pop(stack);
      stack.push_back(make_point({get_real(x), get_real(y), get_real(z)}));
      break;
    }
    case Op::Getx: { auto p = pop(stack); stack.push_back(make_real(get_point(p)[0])); break; }
    case Op::Gety: { auto p = pop(stack); stack.push_back(make_real(get_point(p)[1])); break; }
    case Op::Getz: { auto p = pop(stack); stack.push_back(make_real(get_point(p)[2])); break; }
    case Op::Get: {
      auto i = pop(stack), a = pop(stack);
      int64_t iv = get_integer(i);
      if (a->kind != Value::Array) throw GmlError("WrongTypeRef");
      const int64_t len = (int64_t)a->arr.size();
      if (iv < 0 || iv > len) throw GmlError("ArrayOutOfBounds");
      if (iv == len) throw GmlError("index out of bounds");  // src/evaluator.rs:549-553 panics here
      stack.push_back(a->arr[(size_t)(len - iv - 1)]);
      break;
    }
    case Op::Length: {
      auto a = pop(stack);
      if (a->kind != Value::Array) throw GmlError("WrongTypeRef");
      stack.push_back(make_int((int64_t)a->arr.size()));
      break;
    }
    case Op::If: {
      auto c2 = pop(stack), c1 = pop(stack), pred = pop(stack);
      const ValuePtr& c = get_boolean(pred) ? c1 : c2;
      if (c->kind != Value::Closure) throw GmlError("WrongType");
      Env e(c->env);
      do_evaluate(e, stack, *c->ast);
      break;
    }
    case Op::Apply: {
      auto c = pop(stack);
      if (c->kind != Value::Closure) throw GmlError("WrongType");
      Env e(c->env);
      do_evaluate(e, stack, *c->ast);
      break;
    }
    case Op::Sphere: push_prim(stack, PrimType::Sphere); break;
    case Op::Cube: push_prim(stack, PrimType::Cube); break;
    case Op::Cylinder: push_prim(stack, PrimType::Cylinder); break;
    case Op::Cone: push_prim(stack, PrimType::Cone); break;
    case Op::Plane: push_prim(stack, PrimType::Plane); break;
    case Op::Union: push_op(stack, Rule::Union); break;
    case Op::Intersect: push_op(stack, Rule::Intersect); break;
    case Op::Difference: push_op(stack, Rule::Difference); break;
    case Op::Translate: push_xform(stack, XformKind::Translate, 3); break;
    case Op::Scale: push_xform(stack, XformKind::Scale, 3); break;
    case Op::Uscale: push_xform(stack, XformKind::Uscale, 1); break;
    case Op::Rotatex: push_xform(stack, XformKind::Rotatex, 1); break;
    case Op::Rotatey: push_xform(stack, XformKind::Rotatey, 1); break;
    case Op::Rotatez: push_xform(stack, XformKind::Rotatez, 1); break;
    case Op::Light: {
      auto color = pop(stack), d = pop(stack);
      auto l = std::make_shared<Light>();
      l->kind = Light::Directional;
      const Vec3& dv = get_point(d);
      l->dir = normalize({-dv[0], -dv[1], -dv[2]});
      l->color = get_point(color);
      auto v = std::make_shared<Value>(); v->kind = Value::LightV; v->light = l;
      stack.push_back(v);
      break;
    }
    case Op::Pointlight: {
      auto color = pop(stack), p = pop(stack);
      auto l = std::make_shared<Light>();
      l->kind = Light::Point;
      l->pos = get_point(p);
      l->color = get_point(color);
      auto v = std::make_shared<Value>(); v->kind = Value::LightV; v->light = l;
      stack.push_back(v);
      break;
    }
    case Op::Spotlight: {
      auto exp = pop(stack), cutoff = pop(stack), color = pop(stack), at = pop(stack), pos = pop(stack);
      auto l = std::make_shared<Light>();
      l->kind = Light::Spot;
      l->pos = get_point(pos);
      const Vec3& atv = get_point(at);
      l->color = get_point(color);
      const double cut = get_real(cutoff);
      l->exp = get_real(exp);
      l->dir = normalize({atv[0] - l->pos[0], atv[1] - l->pos[1], atv[2] - l->pos[2]});
      l->coscutoff = std::cos(to_radians(cut));
      auto v = std::make_shared<Value>(); v->kind = Value::LightV; v->light = l;
      stack.push_back(v);
      break;
    }
    case Op::Render: {
      auto file = pop(stack), ht = pop(stack), wid = pop(stack), fov = pop(stack), depth = pop(stack),
           obj = pop(stack), lights = pop(stack), amb = pop(stack);
      RenderCall rc;
      rc.amb = get_point(amb);
      if (lights->kind != Value::Array) throw GmlError("WrongTypeRef");
      // the array is stored reversed and consumed in storage order (src/evaluator.rs:531-542)
      for (auto& lv : lights->arr) {
        if (lv->kind != Value::LightV) throw GmlError("WrongTypeRef");
        rc.lights.push_back(lv->light);
      }
      rc.scene = get_node(obj);
      rc.depth = get_integer(depth);
      rc.fov = get_real(fov);
      rc.w = get_integer(wid);
      rc.h = get_integer(ht);
      if (file->kind != Value::String) throw GmlError("WrongTypeRef");
      rc.file = file->s;
      if (hook_) hook_(rc);
      break;
    }
  }
}

void Evaluator::do_evaluate(Env& env, Stack& stack, const AstList& ast) {
  for (const AstNode& item : ast) {
    switch (item.kind) {
      case AstNode::Function: {
        auto v = std::make_shared<Value>();
        v->kind = Value::Closure;
        v->env = env.snapshot();
        v->ast = item.children;
        stack.push_back(std::move(v));
        break;
      }
      case AstNode::Array: {
        Stack local;
        do_evaluate(env, local, *item.children);
        auto v = std::make_shared<Value>();
        v->kind = Value::Array;
        v->arr.assign(local.rbegin(), local.rend());
        stack.push_back(std::move(v));
        break;
      }
      case AstNode::Leaf: {
        const Token& t = item.tok;
        switch (t.kind) {
          case Token::Integer: stack.push_back(make_int(t.i)); break;
          case Token::Real: stack.push_back(make_real(t.r)); break;
          case Token::Boolean: stack.push_back(make_bool(t.b)); break;
          case Token::Str: {
            auto v = std::make_shared<Value>(); v->kind = Value::String; v->s = t.s;
            stack.push_back(std::move(v));
            break;
          }
          case Token::Binder: env.insert(t.s, pop(stack)); break;
          case Token::Identifier: {
            const ValuePtr* v = env.get(t.s);
            if (!v) throw GmlError("environment missing identifier: " + t.s);
            stack.push_back(*v);
            break;
          }
          case Token::Operator: eval_op(t.op, stack); break;
          default: throw GmlError("InvalidAst");
        }
        break;
      }
    }
  }
}

void Evaluator::run(const std::string& text, Env& env, Stack& stack) {
  auto ast = parse(tokenize(text));
  do_evaluate(env, stack, *ast);
}

void Evaluator::call_surface(const Value& closure, int64_t face, double u, double v, Vec3* color,
                             double* kd, double* ks, double* n) {
  Env e(closure.env);
  Stack st;
  st.push_back(make_int(face));
  st.push_back(make_real(u));
  st.push_back(make_real(v));
  do_evaluate(e, st, *closure.ast);
  ValuePtr vn = pop(st), vks = pop(st), vkd = pop(st), vc = pop(st);
  *kd = get_real(vkd);
  *ks = get_real(vks);
  *n = get_real(vn);
  *color = get_point(vc);
}

// ============================================================== preprocessor
namespace {
bool starts_with(const std::string& s, const char* p) { return s.compare(0, std::strlen(p), p) == 0; }

void get_contents(const std::string& path, std::unordered_set<std::string>& defines,
                  std::vector<std::string>& contents) {
  std::ifstream in(path, std::ios::binary);
  if (!in) throw GmlError("cannot open " + path);
  int guard = 0;
  std::string line;
  while (std::getline(in, line)) {
    if (!line.empty() && line.back() == '\r') line.pop_back();
    if (starts_with(line, "#include \"")) {
      std::string base;
      size_t slash = path.find_last_of('/');
      if (slash != std::string::npos) base = path.substr(0, slash + 1);
      if (line.size() < 11) throw GmlError("bad #include line");
      get_contents(base + line.substr(10, line.size() - 11), defines, contents);
    } else if (starts_with(line, "#ifndef ")) {
      if (defines.count(line.substr(8))) guard += 1;
    } else if (starts_with(line, "#define ")) {
      defines.insert(line.substr(8));
    } else if (starts_with(line, "#endif ")) {
      guard -= 1;
    } else if (guard == 0) {
      contents.push_back(line);
    }
  }
}
}  // namespace

std::string preprocess_file(const std::string& path) {
  std::unordered_set<std::string> defines;
  std::vector<std::string> lines;
  get_contents(path, defines, lines);
  std::string out;
  for (size_t i = 0; i < lines.size(); i++) {
    if (i) out += '\n';
    out += lines[i];
  }
  return out;
}

// ============================================================== big-stack runner
namespace {
struct Thunk { const std::function<void()>* fn; std::exception_ptr err; };
void* thunk_main(void* p) {
  Thunk* t = (Thunk*)p;
  try { (*t->fn)(); } catch (...) { t->err = std::current_exception(); }
  return nullptr;
}
}  // namespace

void run_with_big_stack(const std::function<void()>& fn, size_t stack_bytes) {
  pthread_attr_t attr;
  pthread_attr_init(&attr);
  Thunk t{&fn, nullptr};
  pthread_t th;
  int rc = pthread_attr_setstacksize(&attr, stack_bytes);
  if (rc == 0) rc = pthread_create(&th, &attr, thunk_main, &t);
  while (rc != 0 && stack_bytes > ((size_t)64 << 20)) {  // shrink until the kernel grants it
    stack_bytes /= 2;
    pthread_attr_setstacksize(&attr, stack_bytes);
    rc = pthread_create(&th, &attr, thunk_main, &t);
  }
  pthread_attr_destroy(&attr);
  if (rc != 0) throw GmlError("cannot create evaluator thread");
  pthread_join(th, nullptr);
  if (t.err) std::rethrow_exception(t.err);
}

}  // namespace gml
